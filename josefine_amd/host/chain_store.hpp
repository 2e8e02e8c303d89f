// chain_store.hpp — what the reference keeps in sled, byte for byte: ONE ordered
// byte-key tree per partition; key = BlockId bytes (8-byte big-endian id, chain.rs:63-67) ->
// bincode(Block{id, next, data}) (bincode 1.x defaults: little-endian fixed-width integers, u64 length
// prefixes; BlockId goes through serialize_bytes, chain.rs:48-53: length prefix 8 + the 8 id bytes); key
// "commit" -> the 8 big-endian bytes of the commit id (chain.rs:198).  The host mirror's BlockStore IS this
// (raft_handle.hpp); the engine itself only knows ids and parent pointers.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "types.hpp"

namespace josefine {
namespace formats {

using Bytes = std::string;  // raw bytes

struct FormatError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---- bincode (chain store values) ---------------------------------------------------------------
inline void put_u64_le(Bytes& o, uint64_t v) {
  for (int i = 0; i < 8; i++) o.push_back((char)((v >> (8 * i)) & 0xff));
}
inline Bytes block_key(BlockId id) {  // BlockId::new, chain.rs:63-67 — also the sled key
  Bytes k(8, '\0');
  for (int i = 0; i < 8; i++) k[i] = (char)((id >> (8 * (7 - i))) & 0xff);
  return k;
}
inline BlockId key_block_id(const Bytes& k) {
  if (k.size() != 8) throw FormatError("block key is not 8 bytes");
  BlockId id = 0;
  for (int i = 0; i < 8; i++) id = (id << 8) | (uint8_t)k[i];
  return id;
}
inline Bytes encode_block_id(BlockId id) {  // bincode::serialize(&BlockId): chain.rs:346-350
  Bytes o;
  put_u64_le(o, 8);
  o += block_key(id);
  return o;
}
inline Bytes encode_block(const Block& b) {  // bincode::serialize(&block), chain.rs:149,168,187
  Bytes o = encode_block_id(b.id);
  o += encode_block_id(b.next);
  put_u64_le(o, b.data.size());
  o.append((const char*)b.data.data(), b.data.size());
  return o;
}
struct Reader {
  const Bytes& s;
  size_t at = 0;
  uint64_t u64_le() {
    if (at + 8 > s.size()) throw FormatError("bincode: unexpected end of input");
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)(uint8_t)s[at + i] << (8 * i);
    at += 8;
    return v;
  }
  Bytes bytes() {
    const uint64_t n = u64_le();
    if (n > s.size() - at) throw FormatError("bincode: length prefix runs past the end of input");
    Bytes b = s.substr(at, n);
    at += n;
    return b;
  }
};
inline BlockId decode_block_id(const Bytes& v) {
  Reader r{v};
  return key_block_id(r.bytes());
}
// bincode::deserialize::<Block> — fails on the "commit" key's 8-byte value exactly as the
// reference does (chain.rs:219-226: the length prefix it reads is the big-endian id seen
// little-endian: astronomically large)
inline Block decode_block(const Bytes& v) {
  Reader r{v};
  Block b;
  b.id = key_block_id(r.bytes());
  b.next = key_block_id(r.bytes());
  const Bytes d = r.bytes();
  b.data.assign(d.begin(), d.end());
  return b;
}

// The chain's sled tree as the reference lays it out: ONE ordered byte-key map holding the blocks
// and the "commit" key.  What a drop-in keeps on disk next to the engine (the engine itself only
// knows ids and parent pointers).
class ChainStore {
 public:
  static const Bytes& commit_key() {
    static const Bytes k = "commit";  // chain.rs:120,198
    return k;
  }
  void insert(const Block& b) { kv_[block_key(b.id)] = encode_block(b); }  // upsert, chain.rs:149,168,187
  bool has(BlockId id) const { return kv_.count(block_key(id)) != 0; }     // chain.rs:155-157
  void remove(BlockId id) { kv_.erase(block_key(id)); }                    // chain.rs:246
  void set_commit(BlockId id) { kv_[commit_key()] = block_key(id); }       // chain.rs:198
  uint64_t commit() const {                                                // chain.rs:119-123
    auto it = kv_.find(commit_key());
    return it == kv_.end() ? 0 : key_block_id(it->second);
  }
  Block get(BlockId id) const {
    auto it = kv_.find(block_key(id));
    if (it == kv_.end()) throw FormatError("no such block");
    return decode_block(it->second);
  }
  // Chain::range(lo..) / (lo..hi) / (lo..=hi): blocks in key order, at most `limit` of them.  An
  // unbounded range runs into the "commit" key once the blocks are exhausted and dies there, as
  // the reference's iterator does (chain.rs:219-226) — unless the caller has
  // stopped pulling before (`limit`).
  std::vector<Block> range(BlockId lo, const BlockId* hi, bool hi_inclusive, size_t limit = SIZE_MAX) const {
    std::vector<Block> out;
    const Bytes hk = hi ? block_key(*hi) : Bytes();
    for (auto it = kv_.lower_bound(block_key(lo)); it != kv_.end() && out.size() < limit; ++it) {
      if (hi && (it->first > hk || (it->first == hk && !hi_inclusive))) break;
      out.push_back(decode_block(it->second));  // throws on the commit key's value
    }
    return out;
  }
  size_t entries() const { return kv_.size(); }
  const std::map<Bytes, Bytes>& raw() const { return kv_; }
  // map-like conveniences of the host mirror
  size_t count(BlockId id) const { return has(id) ? 1 : 0; }
  Block at(BlockId id) const { return get(id); }
  // the largest block key below `id` (0 if there is none): what Chain::append's `next = head` is for a
  // chain built by appends only (chain.rs:164-167)
  BlockId prev_key(BlockId id) const {
    auto it = kv_.lower_bound(block_key(id));
    while (it != kv_.begin()) {
      --it;
      if (it->first.size() == 8) return key_block_id(it->first);
    }
    return 0;
  }
  // sled re-opened: the same bytes in a fresh tree (what a process restart finds on disk)
  static ChainStore from_raw(const std::map<Bytes, Bytes>& kv) {
    ChainStore s;
    s.kv_ = kv;
    return s;
  }
  // Chain::new on this tree (chain.rs:117-137): commit = the "commit" key or 0; head = id_gen = commit (Q8: NOT
  // the last stored block); an empty tree gets the genesis block and id_gen 1
  struct Reopened {
    BlockId commit, head, id_gen;
  };
  Reopened reopen() {
    const BlockId c = commit();
    if (c == 0) {
      insert(Block{0, 0, {}});  // init(): genesis (chain.rs:139-153)
      return Reopened{0, 0, 1};
    }
    return Reopened{c, c, c};
  }

 private:
  std::map<Bytes, Bytes> kv_;  // byte-wise key order = sled's
};

}  // namespace formats
}  // namespace josefine
