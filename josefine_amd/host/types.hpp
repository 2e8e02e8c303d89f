// types.hpp — the reference's vocabulary for the C++ host mirror: NodeId / Term / BlockId, Block, Command,
// Address, Message, Instruction (src/raft/mod.rs:136-141,160-227; chain.rs:86-91; rpc.rs:5-27; fsm.rs:20-29).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/josefine_gpu.h"

namespace josefine {

using NodeId = uint32_t;   // mod.rs:136
using Term = uint64_t;     // mod.rs:139
using BlockId = uint64_t;  // chain.rs:29-36 (8-byte BE id, numeric order)

struct Block {  // chain.rs:86-91
  BlockId id = 0, next = 0;
  std::vector<uint8_t> data;
};

struct Command {  // mod.rs:160-227
  uint8_t kind = JG_CMD_NOOP;
  NodeId from = 0;
  Term term = 0;
  uint64_t id = 0, aux = 0;
  bool flag = false;
  std::vector<Block> blocks;      // AppendEntries
  std::vector<uint8_t> proposal;  // ClientRequest payload (rpc.rs:30-40)

  static Command Tick() { return mk(JG_CMD_TICK); }
  static Command Propose() { return mk(JG_CMD_PROPOSE); }
  static Command Timeout() { return mk(JG_CMD_TIMEOUT); }
  static Command Noop() { return mk(JG_CMD_NOOP); }
  static Command VoteRequest(Term term, NodeId candidate_id, Term last_term, BlockId head) {
    Command c = mk(JG_CMD_VOTE_REQUEST);
    c.term = term, c.from = candidate_id, c.aux = last_term, c.id = head;
    return c;
  }
  static Command VoteResponse(Term term, NodeId from, bool granted) {
    Command c = mk(JG_CMD_VOTE_RESPONSE);
    c.term = term, c.from = from, c.flag = granted;
    return c;
  }
  static Command AppendEntries(Term term, NodeId leader_id, std::vector<Block> blocks) {
    Command c = mk(JG_CMD_APPEND_ENTRIES);
    c.term = term, c.from = leader_id, c.blocks = std::move(blocks);
    return c;
  }
  static Command AppendResponse(NodeId node_id, Term term, BlockId head, bool success) {
    Command c = mk(JG_CMD_APPEND_RESPONSE);
    c.from = node_id, c.term = term, c.id = head, c.flag = success;
    return c;
  }
  static Command Heartbeat(Term term, BlockId commit, NodeId leader_id) {
    Command c = mk(JG_CMD_HEARTBEAT);
    c.term = term, c.id = commit, c.from = leader_id;
    return c;
  }
  static Command HeartbeatResponse(BlockId commit, bool has_committed) {
    Command c = mk(JG_CMD_HEARTBEAT_RESPONSE);
    c.id = commit, c.flag = has_committed;
    return c;
  }
  static Command ClientRequest(uint64_t request_id, std::vector<uint8_t> proposal) {
    Command c = mk(JG_CMD_CLIENT_REQUEST);
    c.id = request_id, c.proposal = std::move(proposal);
    return c;
  }
  static Command ClientResponse(uint64_t request_id) {
    Command c = mk(JG_CMD_CLIENT_RESPONSE);
    c.id = request_id;
    return c;
  }

 private:
  static Command mk(uint8_t k) {
    Command c;
    c.kind = k;
    return c;
  }
};

struct Address {  // rpc.rs:5-14
  uint8_t kind = JG_TO_PEERS;
  NodeId peer = 0;
};
struct Message {  // rpc.rs:17-27
  uint32_t group = 0;
  Address from, to;
  Command command;
};
struct Instruction {  // fsm.rs:20-29
  enum Kind { Apply, Notify } kind = Apply;
  uint32_t group = 0;
  Block block;              // Apply
  uint64_t request_id = 0;  // Notify
  BlockId block_id = 0;     // Notify
};

class EngineError : public std::runtime_error {  // anyhow::Error
 public:
  EngineError(int status, const char* msg) : std::runtime_error(std::string("josefine engine: ") + msg), status(status) {}
  int status;
};

}  // namespace josefine
