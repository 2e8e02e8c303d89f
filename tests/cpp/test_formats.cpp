// Known-answer tests of josefine_amd/host/formats.hpp (host-only: g++, no GPU, no engine library).
//   chain.rs:345-350  block_id_serde            bincode(BlockId::new(0)) round trip
//   tcp.rs:172-196    read_message              frame + serde_json of Message{Peer(1), Peer(2), Tick}
//   tcp.rs:198-232    send_message              the same frame, read back
// plus one vector per Command variant, hand-derived from the serde attributes in mod.rs / rpc.rs /
// chain.rs (externally tagged enums, declaration order, BlockId = array of its 8 big-endian bytes),
// the sled key order of the chain store, and the reference's own failure on the "commit" key.
#include <cassert>
#include <cstdio>

#include "../../josefine_amd/host/formats.hpp"

using namespace josefine;
using namespace josefine::formats;

static Bytes B(std::initializer_list<int> v) {
  Bytes b;
  for (int x : v) b.push_back((char)x);
  return b;
}
static Message msg(Address from, Address to, Command c) {
  Message m;
  m.from = from, m.to = to, m.command = std::move(c);
  return m;
}
static Address peer(NodeId id) {
  Address a;
  a.kind = JG_TO_PEER, a.peer = id;
  return a;
}
static Address peers() {
  Address a;
  a.kind = JG_TO_PEERS;
  return a;
}
static void same(const Command& a, const Command& b) {
  assert(a.kind == b.kind && a.from == b.from && a.term == b.term && a.id == b.id && a.aux == b.aux && a.flag == b.flag);
  assert(a.proposal == b.proposal && a.blocks.size() == b.blocks.size());
  for (size_t i = 0; i < a.blocks.size(); i++)
    assert(a.blocks[i].id == b.blocks[i].id && a.blocks[i].next == b.blocks[i].next && a.blocks[i].data == b.blocks[i].data);
}
static void roundtrip(const Message& m, const char* expect_json) {
  const std::string j = encode_message(m);
  if (expect_json && j != expect_json) {
    std::fprintf(stderr, "got  %s\nwant %s\n", j.c_str(), expect_json);
    assert(false);
  }
  const Message back = decode_message(j);
  assert(back.from.kind == m.from.kind && back.from.peer == m.from.peer && back.to.kind == m.to.kind && back.to.peer == m.to.peer);
  same(back.command, m.command);
}

int main() {
  // ---- chain.rs:345-350 block_id_serde
  assert(encode_block_id(0) == B({8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}));
  assert(decode_block_id(encode_block_id(0)) == 0);
  assert(encode_block_id(0x0102030405060708ull) == B({8, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8}));
  assert(decode_block_id(encode_block_id(666)) == 666);
  // Block{id 2, next 1, data [0xAA, 0xBB]} as bincode 1.x lays it out
  Block b;
  b.id = 2, b.next = 1, b.data = {0xAA, 0xBB};
  assert(encode_block(b) == B({8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2,  8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1,
                               2, 0, 0, 0, 0, 0, 0, 0, 0xAA, 0xBB}));
  const Block back = decode_block(encode_block(b));
  assert(back.id == 2 && back.next == 1 && back.data == b.data);
  // genesis (chain.rs:139-153): id 0, next 0, no data -> 40 bytes
  Block g;
  assert(encode_block(g).size() == 40);

  // ---- the sled tree: block keys in numeric order, the "commit" key behind them; an unbounded range dies
  //      on it (chain.rs:219-226, Q9), a bounded one or one that stops pulling in time does not
  ChainStore st;
  st.insert(g);
  for (BlockId i = 1; i <= 300; i++) {
    Block x;
    x.id = i, x.next = i - 1, x.data = {(uint8_t)i};
    st.insert(x);
  }
  assert(st.commit() == 0 && st.has(256) && !st.has(301));
  assert(st.range(255, nullptr, false).size() == 46);  // 255..300: byte order == numeric order across the 0xff/0x100 edge
  st.set_commit(7);
  assert(st.commit() == 7 && st.entries() == 302);
  assert(st.raw().rbegin()->first == ChainStore::commit_key() && st.raw().rbegin()->second == block_key(7));
  bool died = false;
  try {
    st.range(298, nullptr, false);
  } catch (const FormatError&) {
    died = true;
  }
  assert(died);
  assert(st.range(290, nullptr, false, 6).size() == 6);  // skip(1).take(5) with enough blocks: never reaches the key
  const BlockId hi = 7;
  assert(st.range(3, &hi, true).size() == 5 && st.range(3, &hi, false).size() == 4);  // leader.rs:93 / follower.rs:204
  died = false;
  try {
    decode_block(st.raw().rbegin()->second);
  } catch (const FormatError&) {
    died = true;
  }
  assert(died);
  st.remove(4);  // compact() (chain.rs:246)
  assert(!st.has(4) && st.get(5).next == 4);

  // ---- tcp.rs:172-196 / 198-232: the one message the reference's tests put on the wire
  const Message tick = msg(peer(1), peer(2), Command::Tick());
  const std::string tick_json = "{\"from\":{\"Peer\":1},\"to\":{\"Peer\":2},\"command\":\"Tick\"}";
  assert(encode_message(tick) == tick_json);
  Bytes wire = frame(tick_json);
  assert(wire.size() == 4 + tick_json.size() && wire.substr(0, 4) == B({0, 0, 0, (int)tick_json.size()}));
  Bytes rx = wire.substr(0, 10);  // a partial frame is not a frame yet
  std::string payload;
  assert(!unframe(rx, &payload));
  rx += wire.substr(10) + wire;  // the rest, and a second frame right behind it
  assert(unframe(rx, &payload) && payload == tick_json && rx.size() == wire.size());
  assert(decode_message(payload).command.kind == JG_CMD_TICK);
  assert(unframe(rx, &payload) && rx.empty());

  // ---- one vector per Command variant (serde: externally tagged, declaration order)
  roundtrip(msg(peer(3), peers(), Command::VoteRequest(7, 3, 6, 9)),
            "{\"from\":{\"Peer\":3},\"to\":\"Peers\",\"command\":{\"VoteRequest\":{\"term\":7,\"candidate_id\":3,\"last_term\":6,"
            "\"head\":[0,0,0,0,0,0,0,9]}}}");
  roundtrip(msg(peer(2), peer(3), Command::VoteResponse(7, 2, true)),
            "{\"from\":{\"Peer\":2},\"to\":{\"Peer\":3},\"command\":{\"VoteResponse\":{\"term\":7,\"from\":2,\"granted\":true}}}");
  Block b1, b2;
  b1.id = 257, b1.next = 256, b1.data = {1, 2, 3};
  b2.id = 258, b2.next = 257;
  roundtrip(msg(peer(1), peer(2), Command::AppendEntries(4, 1, {b1, b2})),
            "{\"from\":{\"Peer\":1},\"to\":{\"Peer\":2},\"command\":{\"AppendEntries\":{\"term\":4,\"leader_id\":1,\"blocks\":["
            "{\"id\":[0,0,0,0,0,0,1,1],\"next\":[0,0,0,0,0,0,1,0],\"data\":[1,2,3]},"
            "{\"id\":[0,0,0,0,0,0,1,2],\"next\":[0,0,0,0,0,0,1,1],\"data\":[]}]}}}");
  roundtrip(msg(peer(1), peer(2), Command::AppendEntries(4, 1, {})), nullptr);
  roundtrip(msg(peer(2), peer(1), Command::AppendResponse(2, 4, 258, true)),
            "{\"from\":{\"Peer\":2},\"to\":{\"Peer\":1},\"command\":{\"AppendResponse\":{\"node_id\":2,\"term\":4,"
            "\"head\":[0,0,0,0,0,0,1,2],\"success\":true}}}");
  roundtrip(msg(peer(1), peers(), Command::Heartbeat(4, 200, 1)),
            "{\"from\":{\"Peer\":1},\"to\":\"Peers\",\"command\":{\"Heartbeat\":{\"term\":4,\"commit\":[0,0,0,0,0,0,0,200],\"leader_id\":1}}}");
  roundtrip(msg(peer(2), peer(1), Command::HeartbeatResponse(199, false)),
            "{\"from\":{\"Peer\":2},\"to\":{\"Peer\":1},\"command\":{\"HeartbeatResponse\":{\"commit\":[0,0,0,0,0,0,0,199],"
            "\"has_committed\":false}}}");
  Command cr = Command::ClientRequest(0x0123456789abcdefull, {104, 105});
  cr.from = 2;  // address rewritten to the forwarding follower (follower.rs:260)
  roundtrip(msg(peer(2), peer(1), cr),
            "{\"from\":{\"Peer\":2},\"to\":{\"Peer\":1},\"command\":{\"ClientRequest\":{\"id\":\"00000000-0000-0000-0123-456789abcdef\","
            "\"address\":{\"Peer\":2},\"proposal\":[104,105]}}}");
  Command ok = Command::ClientResponse(5);
  ok.proposal = {7};
  Address client;
  client.kind = JG_TO_CLIENT;
  Address local;
  local.kind = JG_TO_LOCAL;
  roundtrip(msg(local, client, ok),
            "{\"from\":\"Local\",\"to\":\"Client\",\"command\":{\"ClientResponse\":{\"id\":\"00000000-0000-0000-0000-000000000005\","
            "\"res\":{\"Ok\":[7]}}}}");
  Command err = Command::ClientResponse(5);
  err.flag = true;
  roundtrip(msg(local, client, err), nullptr);
  for (Command c : {Command::Propose(), Command::Timeout(), Command::Noop()}) roundtrip(msg(peer(1), peer(1), c), nullptr);

  // malformed input is an error, not garbage
  for (const char* bad : {"", "{", "{\"from\":\"Nobody\",\"to\":\"Peers\",\"command\":\"Tick\"}",
                          "{\"from\":\"Peers\",\"to\":\"Peers\",\"command\":{\"Heartbeat\":{\"term\":1,\"commit\":[1,2,3],\"leader_id\":1}}}",
                          "{\"from\":\"Peers\",\"to\":\"Peers\",\"command\":\"Tick\"} x"}) {
    bool threw = false;
    try {
      decode_message(bad);
    } catch (const FormatError&) {
      threw = true;
    }
    assert(threw);
  }
  std::puts("formats ok");
  return 0;
}
