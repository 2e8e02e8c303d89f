// jg_votes.h - the ELECTION vocabulary as mailbox words: the receiving half (DESIGN.md "What comes next").
//
// NOT part of the engine yet: nothing in josefine_gpu.hip includes this file.  It holds the per-group logic of the step
// that is to replace the row transport for a routed round's vote traffic, developed against the oracle on the HOST
// (tests/host_compiled.py compiles it with the device's state machine; tests/test_vote_half.py) so that the next round
// spends its GPU-minutes on the integration and the memory system, not on the semantics.  What it says is in
// tests/election_words.py (numpy), held there to the rows the routed clusters really exchange.
//
// One round's inbound vote traffic of ONE node, per sender slot s and partition g:
//   request   sender s campaigns (candidate.rs:24-44): q_n[s][g] identical VoteRequest{term, candidate_id = id(s),
//             last_term = term, head} (q_n = config.nodes.len() = R - 1 copies, Q5; 0: none)
//   answer    sender s answers a campaign of the node a_to[s][g] (follower.rs:219-246, candidate.rs:66-84): a_n[s][g]
//             VoteResponse{from = id(s), term, granted}: the first `first`, every further one `rest`
//   q_at / a_at: where the stretch begins in the sender's run for this partition (a sender that answers and campaigns
//   within one round sends both; the transport's order is (sender slot, emission order))
// The half applies them exactly as the rows would have been applied - per partition in the order (sender slot, emission
// order), one jg_apply per copy, one election_status() per VoteResponse - and emits what the rows would have emitted:
// the VoteResponses this node gives to ONE requester per partition as its own answer word (o_*), everything else (a second
// requester's answers, the Heartbeat of elect(), candidate.rs:108-113) as rows on the exceptional queue with their emission
// index, so that words and rows merge back into the reference's emission order.
#pragma once
#include "jg_device.h"

struct JgVoteIn {  // [R][G] each, indexed [sender slot][partition]
  const uint64_t* q_term;
  const uint64_t* q_head;
  const uint8_t* q_n;   // copies (0: no request from this sender)
  const uint8_t* q_at;
  const uint64_t* a_term;
  const uint8_t* a_n;   // copies (0: no answer from this sender)
  const uint8_t* a_at;
  const uint8_t* a_bits;  // bit 0: the first answer, bit 1: every further one
  const uint8_t* a_to;    // the slot the answers are addressed to
};
struct JgVoteOut {  // [G] each: this node's own answer word of the round
  uint64_t* term;
  uint8_t* n;     // 0: none
  uint8_t* at;    // emission index of the first copy within this node's step for the partition
  uint8_t* bits;
  uint8_t* to;
};

// one partition of one node; returns the number of quorum decisions taken (election_status evaluations)
__device__ inline uint32_t jg_vote_half_group(const JgDev& d, uint32_t g, uint32_t self, const JgVoteIn& in, const JgVoteOut& out,
                                              uint64_t now, uint32_t seq) {
  const uint32_t G = d.G, R = d.R;
  out.n[g] = 0;
  bool any = false;
  for (uint32_t s = 0; s < R; s++) {
    if (s == self) continue;
    const size_t i = (size_t)s * G + g;
    any = any || in.q_n[i] != 0 || (in.a_n[i] != 0 && in.a_to[i] == self);
  }
  if (!any) return 0;
  JgLane L;
  jg_load(d, L, g);
  const JgLane O = L;
  L.now = now;
  L.seq = seq;
  jg_msg_row buf[JG_MAX_REPLICAS + 1];
  jg_fsm_row sink[2];
  uint32_t k_emit = 0;  // emission index within this node's step for the partition
  uint64_t o_term = 0;
  uint32_t o_n = 0, o_at = 0, o_first = 0, o_rest = 0, o_to = 0;
  for (uint32_t s = 0; s < R; s++) {
    if (s == self) continue;
    const size_t i = (size_t)s * G + g;
    const bool has_q = in.q_n[i] != 0, has_a = in.a_n[i] != 0 && in.a_to[i] == self;
    // the sender's two stretches in its own emission order
    for (int pass = 0; pass < 2; pass++) {
      const bool ans_first = has_q && has_a && in.a_at[i] < in.q_at[i];
      const bool do_ans = (pass == 0) == (ans_first || !has_q);
      if (do_ans ? !has_a : !has_q) continue;
      const uint32_t copies = do_ans ? in.a_n[i] : in.q_n[i];
      for (uint32_t c = 0; c < copies; c++) {
        JgCmd cmd;
        cmd.from = d.node_ids[s];
        if (do_ans) {
          cmd.kind = JG_CMD_VOTE_RESPONSE, cmd.term = in.a_term[i], cmd.id = 0, cmd.aux = 0;
          cmd.flag = (in.a_bits[i] >> (c ? 1 : 0)) & 1u;
        } else {
          cmd.kind = JG_CMD_VOTE_REQUEST, cmd.term = in.q_term[i], cmd.id = in.q_head[i], cmd.aux = in.q_term[i], cmd.flag = 0;
        }
        L.mp = buf, L.mend = buf + JG_MAX_REPLICAS + 1;
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
            } else {
              *d.err = 6;
            }
          }
        }
        if (L.overflow) *d.err = 1;
      }
    }
  }
  if (o_n) out.term[g] = o_term, out.n[g] = (uint8_t)o_n, out.at[g] = (uint8_t)o_at, out.bits[g] = (uint8_t)(o_first | o_rest << 1), out.to[g] = (uint8_t)o_to;
  const uint32_t dec = L.decisions;
  jg_store_dirty<false>(d, L, O);
  return dec;
}

// the half as a kernel: one lane per partition (HBM-bound once the words are packed: 2 x 16 B per sender and partition
// read, the cold records of the partitions that have mail)
__global__ __launch_bounds__(JG_BLOCK) void k_vote_half(JgDev d, uint32_t self, JgVoteIn in, JgVoteOut out, uint64_t now, uint32_t seq) {
  uint32_t dec = 0;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < d.G; g += gridDim.x * JG_BLOCK) dec += jg_vote_half_group(d, g, self, in, out, now, seq);
  if (dec) (void)__hip_atomic_fetch_add(&d.blk_decisions[blockIdx.x], (uint64_t)dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
