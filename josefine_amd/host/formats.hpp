// formats.hpp — the byte formats on either side of the hot path, in C++ for
// the host adapter: what the reference keeps in sled and what it puts on the wire.  Host code only;
// nothing here runs on the device, and none of it is Raft logic.
//
//   chain store  (src/raft/chain.rs:117-153,160-205): one sled tree; key = BlockId bytes (8-byte
//                big-endian id, chain.rs:63-67) -> bincode(Block{id, next, data}); key "commit" ->
//                the 8 big-endian bytes of the commit id (chain.rs:198).  bincode 1.x defaults:
//                little-endian fixed-width integers, u64 length prefixes; BlockId goes through
//                serialize_bytes (chain.rs:48-53): length prefix 8 + the 8 id bytes.
//   peer wire    (src/raft/tcp.rs:40-51,143-156): LengthDelimitedCodec frames (4-byte big-endian
//                payload length) around serde_json(Message{from, to, command}), serde's default
//                externally-tagged enums, struct fields in declaration order (rpc.rs:17-27,
//                mod.rs:160-227).  BlockId is a JSON array of its 8 bytes; Uuid request ids are
//                hyphenated lower-case strings (the adapter's 64-bit tokens ride in the low half).
//
// Deviation, on the reading side only: the reference's own BlockId deserializer asks serde for a
// BORROWED byte slice (chain.rs:55-61), which serde_json cannot produce from a JSON array, so a
// josefine peer fails to decode every message that carries a BlockId (VoteRequest, AppendEntries,
// AppendResponse, Heartbeat, HeartbeatResponse) — only Tick-like messages survive its own wire
// (the one tcp.rs test sends Command::Tick).  decode_message() here accepts what encode_message()
// and the reference's serializer write.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "raft_handle.hpp"

namespace josefine {
namespace formats {


// ---- serde_json(Message) + LengthDelimitedCodec (peer wire) -----------------------------------
inline void json_bytes(std::string& o, const uint8_t* p, size_t n) {
  o += '[';
  for (size_t i = 0; i < n; i++) {
    if (i) o += ',';
    o += std::to_string((unsigned)p[i]);
  }
  o += ']';
}
inline void json_block_id(std::string& o, BlockId id) {
  const Bytes k = block_key(id);
  json_bytes(o, (const uint8_t*)k.data(), 8);
}
inline std::string uuid_of(uint64_t token) {  // ClientRequestId = Uuid (mod.rs:142): the token in the low half
  char buf[40];
  std::snprintf(buf, sizeof buf, "00000000-0000-0000-%04x-%012llx", (unsigned)(token >> 48),
                (unsigned long long)(token & 0xffffffffffffull));
  return buf;
}
inline void json_address(std::string& o, const Address& a) {  // rpc.rs:5-14
  switch (a.kind) {
    case JG_TO_PEERS: o += "\"Peers\""; break;
    case JG_TO_PEER: o += "{\"Peer\":" + std::to_string(a.peer) + "}"; break;
    case JG_TO_LOCAL: o += "\"Local\""; break;
    default: o += "\"Client\""; break;
  }
}
inline std::string encode_command(const Command& c) {  // mod.rs:160-227, fields in declaration order
  std::string o;
  auto b = [](bool v) { return v ? "true" : "false"; };
  switch (c.kind) {
    case JG_CMD_TICK: return "\"Tick\"";
    case JG_CMD_PROPOSE: return "\"Propose\"";
    case JG_CMD_TIMEOUT: return "\"Timeout\"";
    case JG_CMD_NOOP: return "\"Noop\"";
    case JG_CMD_VOTE_REQUEST:
      o = "{\"VoteRequest\":{\"term\":" + std::to_string(c.term) + ",\"candidate_id\":" + std::to_string(c.from) +
          ",\"last_term\":" + std::to_string(c.aux) + ",\"head\":";
      json_block_id(o, c.id);
      return o + "}}";
    case JG_CMD_VOTE_RESPONSE:
      return "{\"VoteResponse\":{\"term\":" + std::to_string(c.term) + ",\"from\":" + std::to_string(c.from) +
             ",\"granted\":" + b(c.flag) + "}}";
    case JG_CMD_APPEND_ENTRIES:
      o = "{\"AppendEntries\":{\"term\":" + std::to_string(c.term) + ",\"leader_id\":" + std::to_string(c.from) +
          ",\"blocks\":[";
      for (size_t i = 0; i < c.blocks.size(); i++) {
        if (i) o += ',';
        o += "{\"id\":";
        json_block_id(o, c.blocks[i].id);
        o += ",\"next\":";
        json_block_id(o, c.blocks[i].next);
        o += ",\"data\":";
        json_bytes(o, c.blocks[i].data.data(), c.blocks[i].data.size());
        o += '}';
      }
      return o + "]}}";
    case JG_CMD_APPEND_RESPONSE:
      o = "{\"AppendResponse\":{\"node_id\":" + std::to_string(c.from) + ",\"term\":" + std::to_string(c.term) +
          ",\"head\":";
      json_block_id(o, c.id);
      return o + ",\"success\":" + b(c.flag) + "}}";
    case JG_CMD_HEARTBEAT:
      o = "{\"Heartbeat\":{\"term\":" + std::to_string(c.term) + ",\"commit\":";
      json_block_id(o, c.id);
      return o + ",\"leader_id\":" + std::to_string(c.from) + "}}";
    case JG_CMD_HEARTBEAT_RESPONSE:
      o = "{\"HeartbeatResponse\":{\"commit\":";
      json_block_id(o, c.id);
      return o + ",\"has_committed\":" + b(c.flag) + "}}";
    case JG_CMD_CLIENT_REQUEST: {  // ClientRequest{id, address, proposal} (mod.rs:144-149); the follower
      o = "{\"ClientRequest\":{\"id\":\"" + uuid_of(c.id) + "\",\"address\":";  // rewrites address to itself
      Address a;                                                                    // (follower.rs:260)
      a.kind = JG_TO_PEER, a.peer = c.from;
      json_address(o, a);
      o += ",\"proposal\":";
      json_bytes(o, c.proposal.data(), c.proposal.size());
      return o + "}}";
    }
    case JG_CMD_CLIENT_RESPONSE:  // ClientResponse{id, res: Result<Response, ResponseError>} (mod.rs:151-155)
      o = "{\"ClientResponse\":{\"id\":\"" + uuid_of(c.id) + "\",\"res\":";
      if (c.flag) return o + "{\"Err\":{}}}}";
      o += "{\"Ok\":";
      json_bytes(o, c.proposal.data(), c.proposal.size());
      return o + "}}}";
    default: throw FormatError("not a reference Command");
  }
}
inline std::string encode_message(const Message& m) {  // serde_json::to_string(&Message), rpc.rs:17-27
  std::string o = "{\"from\":";
  json_address(o, m.from);
  o += ",\"to\":";
  json_address(o, m.to);
  return o + ",\"command\":" + encode_command(m.command) + "}";
}
inline Bytes frame(const std::string& payload) {  // LengthDelimitedCodec::new(): 4-byte big-endian length
  Bytes o;
  const uint32_t n = (uint32_t)payload.size();
  for (int i = 3; i >= 0; i--) o.push_back((char)((n >> (8 * i)) & 0xff));
  return o + payload;
}
// one frame off the front of `buf`; false if it is not complete yet
inline bool unframe(Bytes& buf, std::string* payload) {
  if (buf.size() < 4) return false;
  uint32_t n = 0;
  for (int i = 0; i < 4; i++) n = (n << 8) | (uint8_t)buf[i];
  if (n > 8u * 1024 * 1024) throw FormatError("frame exceeds the codec's 8 MiB default");  // LengthDelimitedCodec default
  if (buf.size() < 4 + (size_t)n) return false;
  *payload = buf.substr(4, n);
  buf.erase(0, 4 + (size_t)n);
  return true;
}

// A small JSON reader for exactly this schema (objects, arrays of numbers / objects, strings
// without escapes beyond \" and \\, integers, booleans).
class Json {
 public:
  explicit Json(const std::string& s) : s_(s) {}
  void ws() { while (at_ < s_.size() && (s_[at_] == ' ' || s_[at_] == '\n' || s_[at_] == '\t' || s_[at_] == '\r')) at_++; }
  char peek() { ws(); return at_ < s_.size() ? s_[at_] : '\0'; }
  void expect(char c) {
    if (peek() != c) throw FormatError(std::string("json: expected '") + c + "'");
    at_++;
  }
  bool accept(char c) {
    if (peek() != c) return false;
    at_++;
    return true;
  }
  std::string str() {
    expect('"');
    std::string o;
    while (at_ < s_.size() && s_[at_] != '"') {
      if (s_[at_] == '\\' && at_ + 1 < s_.size()) at_++;
      o += s_[at_++];
    }
    expect('"');
    return o;
  }
  uint64_t num() {
    ws();
    if (at_ >= s_.size() || s_[at_] < '0' || s_[at_] > '9') throw FormatError("json: expected a number");
    uint64_t v = 0;
    while (at_ < s_.size() && s_[at_] >= '0' && s_[at_] <= '9') v = v * 10 + (uint64_t)(s_[at_++] - '0');
    return v;
  }
  bool boolean() {
    ws();
    if (s_.compare(at_, 4, "true") == 0) { at_ += 4; return true; }
    if (s_.compare(at_, 5, "false") == 0) { at_ += 5; return false; }
    throw FormatError("json: expected a boolean");
  }
  std::vector<uint8_t> byte_array() {
    std::vector<uint8_t> o;
    expect('[');
    if (accept(']')) return o;
    do {
      const uint64_t v = num();
      if (v > 255) throw FormatError("json: byte out of range");
      o.push_back((uint8_t)v);
    } while (accept(','));
    expect(']');
    return o;
  }
  BlockId block_id() {
    const std::vector<uint8_t> b = byte_array();
    if (b.size() != 8) throw FormatError("json: BlockId is not 8 bytes");
    BlockId id = 0;
    for (uint8_t x : b) id = (id << 8) | x;
    return id;
  }
  void key(const char* k) {
    if (str() != k) throw FormatError(std::string("json: expected field ") + k);
    expect(':');
  }
  bool done() { ws(); return at_ == s_.size(); }

 private:
  const std::string& s_;
  size_t at_ = 0;
};
inline uint64_t token_of(const std::string& uuid) {
  uint64_t v = 0;
  int digits = 0;
  for (size_t i = uuid.size(); i-- > 0 && digits < 16;) {
    const char c = uuid[i];
    if (c == '-') continue;
    const int d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
    if (d < 0) throw FormatError("json: bad uuid");
    v |= (uint64_t)d << (4 * digits++);
  }
  return v;
}
inline Address decode_address(Json& j) {
  Address a;
  if (j.peek() == '"') {
    const std::string s = j.str();
    a.kind = s == "Peers" ? JG_TO_PEERS : s == "Local" ? JG_TO_LOCAL : s == "Client" ? JG_TO_CLIENT : 0xff;
    if (a.kind == 0xff) throw FormatError("json: unknown Address");
    return a;
  }
  j.expect('{');
  j.key("Peer");
  a.kind = JG_TO_PEER;
  a.peer = (NodeId)j.num();
  j.expect('}');
  return a;
}
inline Command decode_command(Json& j) {
  if (j.peek() == '"') {
    const std::string s = j.str();
    if (s == "Tick") return Command::Tick();
    if (s == "Propose") return Command::Propose();
    if (s == "Timeout") return Command::Timeout();
    if (s == "Noop") return Command::Noop();
    throw FormatError("json: unknown unit Command " + s);
  }
  j.expect('{');
  const std::string v = j.str();
  j.expect(':');
  j.expect('{');
  Command c;
  if (v == "VoteRequest") {
    j.key("term"); const Term term = j.num(); j.expect(',');
    j.key("candidate_id"); const NodeId cand = (NodeId)j.num(); j.expect(',');
    j.key("last_term"); const Term last = j.num(); j.expect(',');
    j.key("head"); const BlockId head = j.block_id();
    c = Command::VoteRequest(term, cand, last, head);
  } else if (v == "VoteResponse") {
    j.key("term"); const Term term = j.num(); j.expect(',');
    j.key("from"); const NodeId from = (NodeId)j.num(); j.expect(',');
    j.key("granted"); const bool g = j.boolean();
    c = Command::VoteResponse(term, from, g);
  } else if (v == "AppendEntries") {
    j.key("term"); const Term term = j.num(); j.expect(',');
    j.key("leader_id"); const NodeId lead = (NodeId)j.num(); j.expect(',');
    j.key("blocks");
    std::vector<Block> blocks;
    j.expect('[');
    if (!j.accept(']')) {
      do {
        Block b;
        j.expect('{');
        j.key("id"); b.id = j.block_id(); j.expect(',');
        j.key("next"); b.next = j.block_id(); j.expect(',');
        j.key("data"); b.data = j.byte_array();
        j.expect('}');
        blocks.push_back(std::move(b));
      } while (j.accept(','));
      j.expect(']');
    }
    c = Command::AppendEntries(term, lead, std::move(blocks));
  } else if (v == "AppendResponse") {
    j.key("node_id"); const NodeId node = (NodeId)j.num(); j.expect(',');
    j.key("term"); const Term term = j.num(); j.expect(',');
    j.key("head"); const BlockId head = j.block_id(); j.expect(',');
    j.key("success"); const bool ok = j.boolean();
    c = Command::AppendResponse(node, term, head, ok);
  } else if (v == "Heartbeat") {
    j.key("term"); const Term term = j.num(); j.expect(',');
    j.key("commit"); const BlockId commit = j.block_id(); j.expect(',');
    j.key("leader_id"); const NodeId lead = (NodeId)j.num();
    c = Command::Heartbeat(term, commit, lead);
  } else if (v == "HeartbeatResponse") {
    j.key("commit"); const BlockId commit = j.block_id(); j.expect(',');
    j.key("has_committed"); const bool has = j.boolean();
    c = Command::HeartbeatResponse(commit, has);
  } else if (v == "ClientRequest") {
    j.key("id"); const uint64_t tok = token_of(j.str()); j.expect(',');
    j.key("address"); const Address a = decode_address(j); j.expect(',');
    j.key("proposal"); std::vector<uint8_t> p = j.byte_array();
    c = Command::ClientRequest(tok, std::move(p));
    c.from = a.kind == JG_TO_PEER ? a.peer : 0;
  } else if (v == "ClientResponse") {
    j.key("id"); const uint64_t tok = token_of(j.str()); j.expect(',');
    j.key("res");
    j.expect('{');
    const std::string r = j.str();
    j.expect(':');
    c = Command::ClientResponse(tok);
    if (r == "Ok") {
      c.proposal = j.byte_array();
    } else {
      j.expect('{');
      j.expect('}');
      c.flag = true;  // Err(ResponseError {})
    }
    j.expect('}');
  } else {
    throw FormatError("json: unknown Command " + v);
  }
  j.expect('}');
  j.expect('}');
  return c;
}
// `as_the_reference_reads`: the documented switch.  The reference's BlockId deserializer asks serde for a
// BORROWED byte slice (chain.rs:55-61); serde_json offers a sequence, so a stock josefine peer fails to
// decode every message that carries a BlockId - VoteRequest, AppendEntries with blocks, AppendResponse,
// Heartbeat, HeartbeatResponse - and only BlockId-free messages (Tick, VoteResponse, ClientRequest, an
// empty AppendEntries ...) survive its own wire.  With the switch on, decode_message fails where it does.
inline bool carries_block_id(const Command& c) {
  switch (c.kind) {
    case JG_CMD_VOTE_REQUEST: case JG_CMD_APPEND_RESPONSE: case JG_CMD_HEARTBEAT: case JG_CMD_HEARTBEAT_RESPONSE: return true;
    case JG_CMD_APPEND_ENTRIES: return !c.blocks.empty();
    default: return false;
  }
}
inline Message decode_message(const std::string& payload, bool as_the_reference_reads = false) {  // serde_json::from_slice::<Message>
  Json j(payload);
  Message m;
  j.expect('{');
  j.key("from"); m.from = decode_address(j); j.expect(',');
  j.key("to"); m.to = decode_address(j); j.expect(',');
  j.key("command"); m.command = decode_command(j);
  j.expect('}');
  if (!j.done()) throw FormatError("json: trailing characters");
  if (as_the_reference_reads && carries_block_id(m.command))
    throw FormatError("json: invalid type: sequence, expected a borrowed byte array (BlockId, chain.rs:55-61)");
  return m;
}

}  // namespace formats
}  // namespace josefine
