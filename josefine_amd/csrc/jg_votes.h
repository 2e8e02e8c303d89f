// jg_votes.h - the ELECTION vocabulary as mailbox words (DESIGN.md "What comes next").
//
// OPT-IN: a routed round uses this only under JG_ROUTE_VOTE_WORDS=1 (josefine_gpu.hip::round_routed_impl); the default
// path does not launch anything in this file.  Built when round 4's GPU-minutes were spent, so held to the oracle on the
// HOST three ways: lane by lane in the device's state machine compiled for the host (tests/host_compiled.py;
// tests/test_vote_half.py, tests/test_vote_mail.py), as kernels on a stand-in with a workgroup's semantics
// (tests/test_host_workgroups.py), and through round_routed_impl itself on an emulated device (tests/test_host_device.py) -
// so that the next GPU-minutes go to the memory system, not to the semantics (profiles/micro/ab_vote_words.sh is the A/B).
// What the words say is in tests/election_words.py (numpy), held there to the rows the routed clusters really exchange.
//
// One round's vote traffic, per SENDER slot s and partition g (JgVoteMail: two of them, a round reads the last one's
// and fills its own):
//   request   sender s campaigns (candidate.rs:24-44): config.nodes.len() = R - 1 identical broadcasts
//             VoteRequest{term, candidate_id = id(s), last_term = term, head} (Q5).  Its rows are emitted as ever (the
//             exceptional queue); the transport's census (jg_votes_census_row) counts the copies into q_ctl and the first
//             one it sees writes (q_term, q_head).  A count other than R - 1 (a second campaign in one round) makes the
//             partition's mail travel as rows for every addressee (k_votes_validate, behind the census).
//   answer    sender s answers a campaign of node `to` (follower.rs:219-246, candidate.rs:66-84): n VoteResponse{from =
//             id(s), term, granted}, the first `first`, every further one `rest`.  Written by the vote half itself
//             (single writer), never rows unless the addressee's partition has to take rows (jg_votes_expand_count / _row).
//   ord       where a stretch begins in the sender's emission order of the round: step << 8 | emission index (a sender
//             that answers and campaigns within one round sends both; the transport's order is (sender slot, step, index))
// and per ADDRESSEE d two bitmaps over the partitions: rowmail[d] - a row that is not such a word is on its way to d
// for g - and wordmail[d] - a word is.  The rule (the same in the delivering pass, the expansion and the receiving half):
// a partition's mail for d travels in words iff EVERYTHING d receives for it this round is such words
// (jg_votes_as_rows is false); otherwise all of it travels as rows, in the transport's order, as without this file.
//
// The receiving half (jg_vote_half_group) applies the words exactly as the rows would have been applied - per partition
// in the order (sender slot, emission order), one jg_apply per copy, one election_status() per VoteResponse - and emits
// what the rows would have emitted: the VoteResponses this node gives to ONE requester per partition as its own answer
// word, everything else (a second requester's answers, the Heartbeat of elect(), candidate.rs:108-113) as rows on the
// exceptional queue with their emission index, so that words and rows merge back into the reference's emission order.
#pragma once
#include "jg_device.h"

#define JG_VOTE_ORD_BITS 11u  // step (3) << 8 | emission index (8)
struct JgVoteMail {
  uint32_t R, G, words;  // words = ceil(G / 64): a bitmap's length
  // per (partition, sender slot), PARTITION-major ([G][R], jg_vote_at): the receiving half reads one partition's words of
  // every sender - R neighbours, one or two 32-byte sectors per column instead of R sectors G entries apart
  uint64_t* q_term;      // valid where q_ctl counts copies
  uint64_t* q_head;
  uint32_t* q_ctl;       // copies (bits 0-7) | the sum of their ords (bits 8-31); clear at the start of a round
  uint64_t* a_term;      // valid where a_ctl says so
  uint32_t* a_ctl;       // n (bits 0-7, 0: none) | ord of the first (8-18) | first (19) | rest (20) | to (21-23); clear at the start of a round
  uint64_t* rowmail;     // [R][words] by addressee; clear at the start of a round
  uint64_t* wordmail;    // [R][words] by addressee; clear at the start of a round
};
__host__ __device__ __forceinline__ size_t jg_vote_at(const JgVoteMail& m, uint32_t sender, uint32_t g) { return (size_t)g * m.R + sender; }
__host__ __device__ inline uint32_t jg_vote_actl(uint32_t n, uint32_t ord, uint32_t first, uint32_t rest, uint32_t to) {
  return n | ord << 8 | (first & 1u) << 19 | (rest & 1u) << 20 | to << 21;
}
// copies that are not a campaign's (Q5: config.nodes.len() of them): `need` = R - 1 in the engine; 0 accepts any count
// (the host tests feed the half words no cluster would send)
__device__ __forceinline__ bool jg_vote_q_ok(uint32_t ctl, uint32_t need) { return !need || (ctl & 0xffu) == need; }

// is this row one copy of a campaign's broadcast, i.e. does it travel in a request word?
__device__ __forceinline__ bool jg_vote_row_is_request_copy(const jg_msg_row& r, uint32_t sender_id, uint32_t k) {
  return r.kind == JG_CMD_VOTE_REQUEST && r.to_kind == JG_TO_PEERS && r.from == sender_id && r.aux == r.term && r.flag == 0 && k < 256u;
}
// the transport's census, once per emitted row of the round (sender slot `src`, the row's step of the round and its
// emission index; `dests`: the members it is addressed to, jg_route_dests) - BEFORE anything is delivered
__device__ inline void jg_votes_census_row(const JgVoteMail& m, uint32_t src, uint32_t sender_id, const jg_msg_row& r, uint32_t step, uint32_t k,
                                           uint32_t dests) {
  if (!dests) return;
  const uint32_t g = r.group;
  const uint64_t bit = 1ull << (g & 63u);
  if (jg_vote_row_is_request_copy(r, sender_id, k)) {
    const size_t i = jg_vote_at(m, src, g);
    const uint32_t old = atomicAdd(&m.q_ctl[i], 1u | ((step & 7u) << 8 | k) << 8);
    if ((old & 0xffu) == 0) {  // (every copy says the same; a second campaign's would not - and is not a word: the count)
      m.q_term[i] = r.term, m.q_head[i] = r.id;
      for (uint32_t b = dests; b; b &= b - 1) atomicOr((unsigned long long*)&m.wordmail[(size_t)(__ffs(b) - 1) * m.words + (g >> 6)], (unsigned long long)bit);
    }
    return;
  }
  for (uint32_t b = dests; b; b &= b - 1) atomicOr((unsigned long long*)&m.rowmail[(size_t)(__ffs(b) - 1) * m.words + (g >> 6)], (unsigned long long)bit);
}
// After the census, once per partition that has word mail: copies that are not a campaign's (Q5: config.nodes.len() of
// them - a second campaign of one sender in a round, or a sender that is no candidate.rs) make the partition's mail
// travel as rows for every addressee of that sender.  (`need` = R - 1; 0: any count is taken - host tests only.)
__device__ inline void jg_votes_validate_group(const JgVoteMail& m, uint32_t g, uint32_t need) {
  for (uint32_t s = 0; s < m.R; s++) {
    const uint32_t c = m.q_ctl[jg_vote_at(m, s, g)];
    if (!(c & 0xffu) || jg_vote_q_ok(c, need)) continue;
    for (uint32_t d = 0; d < m.R; d++)
      if (d != s) atomicOr((unsigned long long*)&m.rowmail[(size_t)d * m.words + (g >> 6)], 1ull << (g & 63u));
  }
}
// does partition g's mail for addressee d travel as rows?  (final once the census and the validation are: one bit)
__device__ __forceinline__ bool jg_votes_as_rows(const JgVoteMail& m, uint32_t d, uint32_t g, uint32_t /*need*/ = 0) {
  return (m.rowmail[(size_t)d * m.words + (g >> 6)] >> (g & 63u)) & 1ull;
}
// the delivering pass: does this row go to addressee d as a row?
__device__ __forceinline__ bool jg_votes_row_travels(const JgVoteMail& m, uint32_t sender_id, const jg_msg_row& r, uint32_t k, uint32_t d, uint32_t need) {
  return !jg_vote_row_is_request_copy(r, sender_id, k) || jg_votes_as_rows(m, d, r.group, need);
}
// sender s's answer word for partition g when its addressee's partition takes rows after all: how many rows it stands
// for (0: none, or the word travels), to whom, and the emission key of the first - row j's is (*step, *k0 + j) ...
__device__ inline uint32_t jg_votes_expand_count(const JgVoteMail& m, uint32_t s, uint32_t g, uint32_t need, uint32_t* to, uint32_t* step, uint32_t* k0) {
  const uint32_t c = m.a_ctl[jg_vote_at(m, s, g)], n = c & 0xffu;
  if (!n) return 0;
  *to = (c >> 21) & 7u;
  if (!jg_votes_as_rows(m, *to, g, need)) return 0;
  const uint32_t ord = (c >> 8) & ((1u << JG_VOTE_ORD_BITS) - 1u);
  *step = ord >> 8, *k0 = ord & 0xffu;
  return n;
}
// ... and row j of them (member_id: the NodeId of every slot)
__device__ inline jg_msg_row jg_votes_expand_row(const JgVoteMail& m, const uint32_t* member_id, uint32_t s, uint32_t g, uint32_t j) {
  const size_t i = jg_vote_at(m, s, g);
  const uint32_t c = m.a_ctl[i];
  jg_msg_row r;
  r.group = g, r.kind = JG_CMD_VOTE_RESPONSE, r.to_kind = JG_TO_PEER, r.pad = 0;
  r.flag = (uint8_t)((c >> (j ? 20 : 19)) & 1u);
  r.to_id = member_id[(c >> 21) & 7u], r.from = member_id[s];
  r.term = m.a_term[i], r.id = 0, r.aux = 0;
  return r;
}

// one partition of one node: `in` is the last round's mail, `out` this round's; returns the number of quorum decisions
// taken (election_status evaluations).  `step`: this step's number within the round (the ord of what it emits)
__device__ inline uint32_t jg_vote_half_group(const JgDev& d, uint32_t g, uint32_t self, const JgVoteMail& in, const JgVoteMail& out, uint32_t need,
                                              uint64_t now, uint32_t seq, uint32_t step) {
  const uint32_t R = d.R;
  if (!((in.wordmail[(size_t)self * in.words + (g >> 6)] >> (g & 63u)) & 1ull)) return 0;
  if (jg_votes_as_rows(in, self, g, need)) return 0;  // (its mail came as rows)
  JgLane L;
  jg_load(d, L, g);
  const JgLane O = L;
  L.now = now;
  L.seq = seq;
  jg_msg_row buf[3];  // (an election's command emits at most two rows - the answer; DROP + the Heartbeat of elect(): prepare_rows' bound for these kinds - and one to spare)
  jg_fsm_row sink[2];
  uint32_t k_emit = 0;  // emission index within this node's step for the partition
  uint64_t o_term = 0;
  uint32_t o_n = 0, o_at = 0, o_first = 0, o_rest = 0, o_to = 0;
  for (uint32_t s = 0; s < R; s++) {
    if (s == self) continue;
    const size_t i = jg_vote_at(in, s, g);
    const uint32_t qc = in.q_ctl[i], ac = in.a_ctl[i];
    const uint32_t q_n = qc & 0xffu, a_n = (ac & 0xffu) && ((ac >> 21) & 7u) == self ? (ac & 0xffu) : 0u;
    if (!q_n && !a_n) continue;
    // the sender's two stretches in its own emission order
    const uint32_t q_ord = q_n ? ((qc >> 8) - q_n * (q_n - 1u) / 2u) / q_n : 0u;  // (the copies' ords are consecutive: candidate.rs:24-44 is one loop)
    const uint32_t a_ord = (ac >> 8) & ((1u << JG_VOTE_ORD_BITS) - 1u);
    const bool ans_first = q_n && a_n && a_ord < q_ord;
    for (int pass = 0; pass < 2; pass++) {
      const bool do_ans = (pass == 0) == (ans_first || !q_n);
      if (do_ans ? !a_n : !q_n) continue;
      const uint32_t copies = do_ans ? a_n : q_n;
      for (uint32_t c = 0; c < copies; c++) {
        JgCmd cmd;
        cmd.from = d.node_ids[s];
        if (do_ans) {
          cmd.kind = JG_CMD_VOTE_RESPONSE, cmd.term = in.a_term[i], cmd.id = 0, cmd.aux = 0;
          cmd.flag = (ac >> (c ? 20 : 19)) & 1u;
        } else {
          cmd.kind = JG_CMD_VOTE_REQUEST, cmd.term = in.q_term[i], cmd.id = in.q_head[i], cmd.aux = in.q_term[i], cmd.flag = 0;
        }
        L.mp = buf, L.mend = buf + 3;
        L.fp = sink, L.fend = sink + 2;
        jg_apply<JG_KINDS_ELECTION>(d, L, cmd, nullptr, nullptr);
        for (const jg_msg_row* r = buf; r != L.mp; r++, k_emit++) {
          const int to = (r->kind == JG_CMD_VOTE_RESPONSE && r->to_kind == JG_TO_PEER) ? jg_slot_of(d, r->to_id) : -1;
          bool folded = false;
          if (to >= 0 && r->id == 0 && r->aux == 0 && k_emit < 256u) {
            if (!o_n) {
              o_term = r->term, o_n = 1, o_at = k_emit, o_first = r->flag, o_rest = 0, o_to = (uint32_t)to, folded = true;
            } else if ((uint32_t)to == o_to && r->term == o_term && k_emit == o_at + o_n && o_n < 255u && (o_n == 1 || r->flag == o_rest)) {
              o_rest = r->flag, o_n++, folded = true;
            }
          }
          if (!folded) {  // a row after all: the exceptional queue, with its place in the emission order
            const uint32_t q = atomicAdd(d.xq_n, 1u);
            if (q < d.xq_cap) {
              JgXqRec x;
              x.row = *r, x.seq = seq, x.k = k_emit;
              d.xq[q] = x;
            }  // (a queue that ran over is seen by the host: xq_n above xq_cap, as for jg_emit_msg's rows)
          }
        }
        if (L.overflow || L.fp != sink) *d.err = 1;  // (an election's command queues nothing for the FSM: nothing is dropped silently)
      }
    }
  }
  if (o_n) {
    const size_t i = jg_vote_at(out, self, g);
    out.a_term[i] = o_term;
    out.a_ctl[i] = jg_vote_actl(o_n, (step & 7u) << 8 | o_at, o_first, o_rest, o_to);
    atomicOr((unsigned long long*)&out.wordmail[(size_t)o_to * out.words + (g >> 6)], 1ull << (g & 63u));
  }
  const uint32_t dec = L.decisions;
  jg_store_dirty<false>(d, L, O);
  return dec;
}

// ---- kernels (one lane per partition; every node of the cluster in one launch: blockIdx.y = node) -------------------
struct JgVoteHalfJob {
  JgDev d;
  uint32_t self, seq, step, need;
  uint64_t now;
};
// the receiving half: a wave skips 64 partitions without mail on one bitmap word (HBM: the bitmaps - G / 8 bytes per node -
// and, per partition with mail, the senders' control words and the partition's cold record)
__global__ __launch_bounds__(JG_BLOCK) void k_vote_half_multi(const JgVoteHalfJob* __restrict__ jobs, JgVoteMail in, JgVoteMail out) {
  const JgVoteHalfJob j = jobs[blockIdx.y];
  uint32_t dec = 0;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < j.d.G; g += gridDim.x * JG_BLOCK) {
    if (!in.wordmail[(size_t)j.self * in.words + (g >> 6)]) continue;  // (wave-uniform: a wave's 64 lanes share the word)
    dec += jg_vote_half_group(j.d, g, j.self, in, out, j.need, j.now, j.seq, j.step);
  }
  if (dec) (void)__hip_atomic_fetch_add(&j.d.blk_decisions[blockIdx.x], (uint64_t)dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the validation (jg_votes_validate_group) of the partitions a wordmail bit names: a wave skips 64 partitions on R words
__global__ __launch_bounds__(JG_BLOCK) void k_votes_validate(JgVoteMail m, uint32_t need) {
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < m.G; g += gridDim.x * JG_BLOCK) {
    uint64_t u = 0;
    for (uint32_t d = 0; d < m.R; d++) u |= m.wordmail[(size_t)d * m.words + (g >> 6)];
    if ((u >> (g & 63u)) & 1ull) jg_votes_validate_group(m, g, need);
  }
}
// a round's mail cleared for its next use - where it was written: a control word is only ever written together with a
// wordmail bit of its partition (the census's first copy, the receiving half's answer), so the union of the addressees'
// wordmail words says which partitions' control words are dirty (R x G / 8 bytes read instead of 8 x R x G bytes written
// per round: 0.6 MB against 40 MB at 1 M x 5); then the bitmaps themselves.  A workgroup owns the partitions of its tile
// and their bitmap words; the barrier keeps a lane from clearing a word its neighbours still have to read.
#if JG_BLOCK % 64 == 0  // (a wave's 64 lanes share a bitmap word; the one-lane host build of the tests has no use for the kernel)
__global__ __launch_bounds__(JG_BLOCK) void k_votes_clear(JgVoteMail m) {
  const uint32_t padded = m.words * 64u;
  for (uint32_t g0 = blockIdx.x * JG_BLOCK; g0 < padded; g0 += gridDim.x * JG_BLOCK) {  // (block-uniform trip count)
    const uint32_t g = g0 + threadIdx.x, w = g >> 6;
    uint64_t u = 0;
    if (w < m.words)
      for (uint32_t d = 0; d < m.R; d++) u |= m.wordmail[(size_t)d * m.words + w];
    __syncthreads();
    if (g < m.G && ((u >> (g & 63u)) & 1ull))
      for (uint32_t s = 0; s < m.R; s++) m.q_ctl[jg_vote_at(m, s, g)] = 0, m.a_ctl[jg_vote_at(m, s, g)] = 0;
    const uint32_t lane = threadIdx.x & 63u;
    if (w < m.words && lane < m.R) m.wordmail[(size_t)lane * m.words + w] = 0, m.rowmail[(size_t)lane * m.words + w] = 0;
  }
}
#endif
