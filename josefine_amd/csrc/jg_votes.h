// jg_votes.h - the ELECTION vocabulary as mailbox words: JG_CLUSTER_OPT_VOTE_WORDS (jg_dense_cluster_set_option), a
// per-cluster option of jg_dense_cluster_round_routed and what bench.py's configs[4] line runs with (DESIGN.md "The vote mail").
// A campaign's R - 1 VoteRequest broadcasts and the VoteResponses they are answered with travel as ONE record per (sender,
// partition) read by a dense receiving half (k_vote_half_multi) instead of as rows through the row transport (jg_route.h);
// what the nodes compute, emit and keep for the host is the row transport's, bit for bit.  Held to the oracle clusters that
// move every message as a row on the device (tests/test_gpu_vote_words.py, tests/test_dense_node.py::test_stationary_*,
// tests/test_gpu_fullsize.py) and, without a GPU, lane by lane in the device's state machine compiled for the host
// (tests/test_vote_half.py, tests/test_vote_mail.py), as kernels on a stand-in with a workgroup's semantics
// (tests/test_host_workgroups.py) and through round_routed_impl itself on an emulated device (tests/test_host_device.py).
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
//             id(s), term, granted}, the first `first`, every further one `rest`, at CONSECUTIVE emission indices.  Written
//             by the vote half itself (single writer), never rows unless the addressee's partition has to take rows
//             (jg_votes_expand_count / _row); an answer that does not continue the word - another requester's, or the same
//             requester's after something else was emitted in between (two campaigns' copies arrive interleaved) - is a
//             row on the exceptional queue, and the census then makes that addressee's partition take rows.
//   ord       where a stretch begins in the sender's emission order of the round: phase << 8 | emission index (a sender
//             that answers and campaigns within one round sends both; the transport's order is (phase, index, sender slot):
//             jg_route.h)
// and per ADDRESSEE d two bitmaps over the partitions: rowmail[d] - a row that is not such a word is on its way to d
// for g - and wordmail[d] - a word is.  The rule (the same in the delivering pass, the expansion and the receiving half):
// a partition's mail for d travels in words iff EVERYTHING d receives for it this round is such words
// (jg_votes_as_rows is false); otherwise all of it travels as rows, in the transport's order, as without this file.
//
// The receiving half (jg_vote_half_group) applies the words exactly as the rows would have been applied - per partition
// in the transport's order: the senders' stretches MERGED by (ord of the copy, sender slot), i.e. every sender's first
// copy before anybody's second (what lets an election of five complete: the candidate sees a quorum of first answers
// before the voters' refusals of the further copies overwrite them, election.rs:33-35), one jg_apply per copy, one
// election_status() per VoteResponse - and emits what the rows would have emitted: the VoteResponses this node gives to
// ONE requester per partition as its own answer word, everything else (a second requester's answers, the Heartbeat of
// elect(), candidate.rs:108-113) as rows on the exceptional queue with their emission index, so that words and rows merge
// back into the reference's emission order.
#pragma once
#include "jg_device.h"
#include "jg_sparse.h"  // jg_block_exclusive_scan

#define JG_VOTE_ORD_BITS 11u  // step (3) << 8 | emission index (8)
// one sender's words for one partition: 32 bytes, so that a partition's mail from all R senders is ONE stretch of 32 R bytes
// (partition-major, jg_vote_at) - as five columns it was five 128-byte lines per partition visit of the receiving half,
// which is bound by exactly those transactions
struct JgVoteRec {
  uint64_t q_term, q_head;  // the request: valid where q_ctl counts copies
  uint64_t a_term;          // the answer: valid where a_ctl says so
  uint32_t q_ctl;           // copies (bits 0-7) | the sum of their ords (bits 8-31); clear at the start of a round
  uint32_t a_ctl;           // n (bits 0-7, 0: none) | ord of the first (8-18) | first (19) | rest (20) | to (21-23); clear at the start of a round
};
static_assert(sizeof(JgVoteRec) == 32, "a sender's words are one 32-byte record");
struct JgVoteMail {
  uint32_t R, G, words, pad;  // words = ceil(G / 64): a bitmap's length
  JgVoteRec* rec;             // [G][R], jg_vote_at
  uint64_t* rowmail;          // [R][words] by addressee; clear at the start of a round
  uint64_t* wordmail;         // [R][words] by addressee; clear at the start of a round
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
// (`need` = R - 1 in the engine: a copy beyond a campaign's makes the partition's mail travel as rows for everybody HERE - what
// k_votes_validate did in a launch of its own until round 6; a sender's copies are all broadcasts, so the copies beyond the
// need-th name exactly the addressees the validation names.  FEWER copies than a campaign's cannot come without a row that
// sets the same bits - a copy whose emission index does not fit the word is a row - and the receiving half says so loudly
// if they ever do.  need = 0: no check here - the host tests validate in a pass of their own, jg_votes_validate_group.)
__device__ __forceinline__ void jg_votes_census_row(const JgVoteMail& m, uint32_t src, uint32_t sender_id, const jg_msg_row& r, uint32_t step, uint32_t k,
                                           uint32_t dests, uint32_t need = 0) {
  if (!dests) return;
  const uint32_t g = r.group;
  const uint64_t bit = 1ull << (g & 63u);
  if (jg_vote_row_is_request_copy(r, sender_id, k)) {
    const size_t i = jg_vote_at(m, src, g);
    const uint32_t old = atomicAdd(&m.rec[i].q_ctl, 1u | ((step & 7u) << 8 | k) << 8);
    if ((old & 0xffu) >= 0x80u || (need && (old & 0xffu) >= need))  // (no campaign has that many copies, and the 8-bit count must not come round to R - 1 again: rows for everybody)
      for (uint32_t b = dests; b; b &= b - 1) atomicOr((unsigned long long*)&m.rowmail[(size_t)(__ffs(b) - 1) * m.words + (g >> 6)], (unsigned long long)bit);
    if ((old & 0xffu) == 0) {  // (every copy says the same; a second campaign's would not - and is not a word: the count)
      m.rec[i].q_term = r.term, m.rec[i].q_head = r.id;
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
    const uint32_t c = m.rec[jg_vote_at(m, s, g)].q_ctl;
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
  const uint32_t c = m.rec[jg_vote_at(m, s, g)].a_ctl, n = c & 0xffu;
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
  const uint32_t c = m.rec[i].a_ctl;
  jg_msg_row r;
  r.group = g, r.kind = JG_CMD_VOTE_RESPONSE, r.to_kind = JG_TO_PEER, r.pad = 0;
  r.flag = (uint8_t)((c >> (j ? 20 : 19)) & 1u);
  r.to_id = 0, r.from = 0;
#pragma unroll
  for (uint32_t k = 0; k < JG_MAX_REPLICAS; k++) {  // (no dynamic index: the caller's table is a by-value copy and would go to scratch)
    if (k == ((c >> 21) & 7u)) r.to_id = member_id[k];
    if (k == s) r.from = member_id[k];
  }
  r.term = m.rec[i].a_term, r.id = 0, r.aux = 0;
  return r;
}

// every field of a lane that a command can change (what jg_store writes): two lanes that agree here are the same replica
__device__ __forceinline__ bool jg_lane_same_state(const JgLane& a, const JgLane& b) {
  return a.term == b.term && a.commit == b.commit && a.head == b.head && a.id_gen == b.id_gen && a.run_hi == b.run_hi &&
         a.election_time == b.election_time && a.heartbeat_time == b.heartbeat_time && a.mword == b.mword && a.mbase == b.mbase &&
         a.flags == b.flags && a.voted_for == b.voted_for && a.leader_id == b.leader_id && a.election_timeout == b.election_timeout &&
         a.rng_draws == b.rng_draws && a.queued == b.queued && a.votes == b.votes;
}
#define JG_KINDS_VOTES ((1u << JG_CMD_VOTE_REQUEST) | (1u << JG_CMD_VOTE_RESPONSE))
// one partition of one node: `in` is the last round's mail, `out` this round's; returns the number of quorum decisions
// taken (election_status evaluations).  `step`: this step's phase of the round (the ord of what it emits).
// The copies are applied in the TRANSPORT's order: merged over the senders by (ord of the copy, sender slot) - a sender's
// request stretch holds the ords q_ord .. q_ord + q_n - 1, its answer stretch a_ord .. a_ord + a_n - 1.  The stretches'
// cursors live in `st` - 2 x JG_MAX_REPLICAS words per lane, `stride` apart: LDS in the kernel (one word per lane and
// stretch, lanes side by side: as register arrays they cost the kernel 45 VGPRs and a wave per SIMD, and a dynamic
// index into registers is scratch), a local array on the host - as next ord | copies left << 12 | copies done << 20
// (| the two answer bits << 28), followed by the senders' answer terms;
// the next copy is the minimum of (next ord, sender, kind) over the unfinished stretches.
// What a copy emits goes where jg_emit_msg's mode 4 puts it: the VoteResponses to one requester, at consecutive emission
// indices, into the lane's answer word, everything else (a second requester's answers, the Heartbeat of elect(),
// candidate.rs:108-113) onto the exceptional queue with its emission index - no row buffer.
// Half of a campaign's applications are not run: where all the stretches of a partition are ALIGNED (the same first ord
// and the same number of copies: a voter's one campaign; a candidate's R - 1 voters, each answering the R - 1 copies in
// its delivered step from index 0) the merged order is level by level - copy c of every stretch, senders ascending - and
// jg_apply is a function of (replica state, command): a level (c >= 1 where answers are read: the first answer may differ
// from the rest) that left the replica as it found it says what every further level does - nothing to the state, the same
// emission, the same decisions - so the rest is accounted for without being run (a voter's 2nd ... R-1st refusal, a
// leader's or a defeated candidate's 2nd ... R-1st round of ignored answers).
#define JG_VOTE_ST_WORDS (4u * JG_MAX_REPLICAS + 4u)  // per lane: 2 R stretch cursors, the R senders' answer terms (two words each), the first campaign's (term, head)
__device__ inline uint32_t jg_vote_half_group(const JgDev& d, uint32_t g, uint32_t self, const JgVoteMail& in, const JgVoteMail& out, uint32_t need,
                                              uint64_t now, uint32_t seq, uint32_t step, uint32_t* st, uint32_t stride) {
  const uint32_t R = d.R;
  if (!((in.wordmail[(size_t)self * in.words + (g >> 6)] >> (g & 63u)) & 1ull)) return 0;
  if (jg_votes_as_rows(in, self, g, need)) return 0;  // (its mail came as rows)
  // A healthy FOLLOWER - four visits of five: the voters - stays one under these two kinds, and all they read or write of
  // it is the flag word, the term, the commit index and the vote record (follower.rs:97-101,219-246: can_vote and
  // apply_vote_request - jg_follower_vote_request / jg_vote_for in jg_device.h must not come to read anything else of a
  // follower, or this path computes on zeros; a VoteResponse at a follower is ignored): four lines instead of the seven
  // jg_load touches, two stores instead of three.
  JgLane L;
  const uint32_t f0 = d.flags[g];
  const bool lean = (f0 & (JGF_ROLE_MASK | JGF_FAULT_MASK)) == JG_ROLE_FOLLOWER;
  if (lean) {
    const uint4 v = d.cold.v[g];
    L.g = g, L.flags = f0, L.term = d.term[g], L.commit = d.commit[g];
    L.voted_for = v.x, L.leader_id = v.y, L.queued = v.z, L.votes = v.w;
    L.head = L.id_gen = L.run_hi = L.heartbeat_time = L.election_time = 0, L.mword = L.mbase = 0;
    L.election_timeout = L.rng_draws = 0;
    L.decisions = 0, L.overflow = 0;
  } else {
    jg_load(d, L, g);
  }
  const JgLane O = L;
  L.now = now;
  L.seq = seq;
  L.xq_on = 4, L.xq_k = 0;  // (xq_k: the emission index within this node's step for the partition)
  L.cap_ack = 0, L.cap_hbc = 0;
  L.mp = L.mend = nullptr;
  L.fp = L.fend = nullptr;  // (an election's command queues nothing for the FSM: a row would raise L.overflow)
  // the senders' stretches: st[2 s] the request's, st[2 s + 1] the answer's; zero where there is none.  Everything a sender
  // said is loaded HERE, R independent 32-byte loads of one stretch of memory: the answers' terms go to the lane's LDS words,
  // the first campaign's (term, head) stay in registers - a voter's one campaign, a candidate's R - 1 answer words then cost
  // the loop below no further trip to memory (one dependent load per copy until round 6: 58 us of a round)
  uint32_t n_st = 0, n_ans = 0, ord0 = 0, cnt0 = 0;
  bool aligned = true;
  uint32_t q1_s = ~0u;  // the first sender with a request word; its payload: four words behind the answer terms
  uint32_t* const at_lo = st + (size_t)(2u * JG_MAX_REPLICAS) * stride;  // a_term of sender s: words [2 s], [2 s + 1]
  uint32_t* const q1 = st + (size_t)(4u * JG_MAX_REPLICAS) * stride;
  for (uint32_t s = 0; s < R; s++) {
    uint32_t wq = 0, wa = 0;
    if (s != self) {
      const uint4* rp = (const uint4*)&in.rec[jg_vote_at(in, s, g)];
      const uint4 lo = rp[0], hi = rp[1];  // {q_term, q_head}, {a_term, q_ctl, a_ctl}
      const uint32_t qc = hi.z, ac = hi.w;
      at_lo[(2u * s) * stride] = hi.x, at_lo[(2u * s + 1u) * stride] = hi.y;
      if ((qc & 0xffu) && q1_s == ~0u) q1_s = s, q1[0] = lo.x, q1[stride] = lo.y, q1[2u * stride] = lo.z, q1[3u * stride] = lo.w;
      const uint32_t q_n = qc & 0xffu, a_n = (ac & 0xffu) && ((ac >> 21) & 7u) == self ? (ac & 0xffu) : 0u;
      if (q_n && !jg_vote_q_ok(qc, need)) *d.err = 1;  // (copies that are not a campaign's and no row bit: the census's rule - see jg_votes_census_row - does not hold)
      if (q_n) {  // (the copies' ords are consecutive - candidate.rs:24-44 is one loop - and q_ctl holds their sum)
        const uint32_t q_ord = ((qc >> 8) - q_n * (q_n - 1u) / 2u) / q_n;
        wq = (q_ord & 0xfffu) | q_n << 12;
        aligned = aligned && (!n_st || (q_ord == ord0 && q_n == cnt0));
        if (!n_st) ord0 = q_ord, cnt0 = q_n;
        n_st++;
      }
      if (a_n) {
        const uint32_t a_ord = (ac >> 8) & ((1u << JG_VOTE_ORD_BITS) - 1u);
        wa = a_ord | a_n << 12 | ((ac >> 19) & 3u) << 28;  // (the answer bits ride in the cursor word: first << 28 | rest << 29)
        aligned = aligned && (!n_st || (a_ord == ord0 && a_n == cnt0));
        if (!n_st) ord0 = a_ord, cnt0 = a_n;
        n_st++, n_ans++;
      }
    }
    st[(2u * s) * stride] = wq, st[(2u * s + 1u) * stride] = wa;
  }
  JgLane P = L;  // (aligned: the replica at the start of the level)
  uint32_t in_level = 0, level = 0;
  for (;;) {
    uint32_t best = ~0u;  // (ord of the stretch's next copy) << 4 | sender << 1 | answer
    for (uint32_t k = 0; k < 2u * R; k++) {
      const uint32_t w = st[k * stride];
      if ((w >> 12) & 0xffu) best = min(best, (w & 0xfffu) << 4 | k);  // (bits 28-29: the answer bits, not part of the key)
    }
    if (best == ~0u) break;
    const uint32_t k = best & 15u, s = k >> 1;
    const bool do_ans = k & 1u;
    const uint32_t w = st[k * stride], c = (w >> 20) & 0xffu;  // (c: this copy's index within its stretch)
    st[k * stride] = w + 1u - (1u << 12) + (1u << 20);
    JgCmd cmd;
    cmd.from = 0;
#pragma unroll
    for (uint32_t r = 0; r < JG_MAX_REPLICAS; r++) cmd.from = s == r ? d.node_ids[r] : cmd.from;  // (a select per slot: an indexed read of the kernel's arguments is a load)
    if (do_ans) {
      cmd.kind = JG_CMD_VOTE_RESPONSE, cmd.term = (uint64_t)at_lo[(2u * s) * stride] | (uint64_t)at_lo[(2u * s + 1u) * stride] << 32;
      cmd.id = 0, cmd.aux = 0, cmd.flag = (w >> (c ? 29 : 28)) & 1u;
    } else {
      cmd.kind = JG_CMD_VOTE_REQUEST, cmd.aux = 0, cmd.flag = 0;
      if (s == q1_s) {
        cmd.term = (uint64_t)q1[0] | (uint64_t)q1[stride] << 32, cmd.id = (uint64_t)q1[2u * stride] | (uint64_t)q1[3u * stride] << 32;
      } else {  // (a second campaign for one partition in one round: its line is in L1)
        const JgVoteRec* rp = &in.rec[jg_vote_at(in, s, g)];
        cmd.term = rp->q_term, cmd.id = rp->q_head;
      }
      cmd.aux = cmd.term;
    }
    jg_apply<JG_KINDS_VOTES>(d, L, cmd, nullptr, nullptr);
    if (!aligned || ++in_level != n_st) continue;
    // a whole level: copy `level` of every stretch
    const uint32_t rem = cnt0 - 1u - level;
    if (rem && (level || !n_ans) && jg_lane_same_state(L, P)) {
      // every further level repeats this one; what it emitted: nothing, or (one stretch) one answer that folded into the word
      const uint32_t rows = L.xq_k - P.xq_k, n = (uint32_t)L.cap_hbc & 0xffu;
      const bool folded = n_st == 1u && rows == 1u && n == ((uint32_t)P.cap_hbc & 0xffu) + 1u && n >= 2u;  // (n >= 2: `rest` is this copy's answer)
      if (!rows || (folded && n + rem <= 255u && L.xq_k + rem <= 256u)) {
        if (folded) L.cap_hbc += rem, L.xq_k += rem;
        L.decisions += rem * (L.decisions - P.decisions);
        break;
      }
    }
    P = L, in_level = 0, level++;
  }
  if ((uint32_t)L.cap_hbc & 0xffu) {
    const size_t i = jg_vote_at(out, self, g);
    const uint32_t w = (uint32_t)L.cap_hbc, to = (w >> 21) & 7u;
    out.rec[i].a_term = L.cap_ack;
    out.rec[i].a_ctl = jg_vote_actl(w & 0xffu, (step & 7u) << 8 | ((w >> 8) & 0xffu), (w >> 19) & 1u, (w >> 20) & 1u, to);
    atomicOr((unsigned long long*)&out.wordmail[(size_t)to * out.words + (g >> 6)], 1ull << (g & 63u));
  }
  if (L.overflow) *d.err = 1;  // (nothing is dropped silently)
  const uint32_t dec = L.decisions;
  if (lean) {
    // (what was not loaded must not have been looked at, let alone changed: a follower that left its role, its term or its
    // commit index under a VoteRequest / VoteResponse would be an engine bug, loud)
    if (jg_role(L) != JG_ROLE_FOLLOWER || L.term != O.term || L.commit != O.commit || L.head | L.id_gen | L.run_hi | L.heartbeat_time | L.election_time |
        L.mword | L.election_timeout | L.rng_draws)
      *d.err = 1;
    if (L.flags != O.flags) d.flags[g] = L.flags;
    if (L.voted_for != O.voted_for || L.leader_id != O.leader_id || L.queued != O.queued || L.votes != O.votes)
      d.cold.v[g] = make_uint4(L.voted_for, L.leader_id, L.queued, L.votes);
  } else {
    jg_store_dirty<false>(d, L, O);
  }
  return dec;
}

// ---- kernels (one lane per partition; every node of the cluster in one launch: blockIdx.y = node) -------------------
struct JgVoteHalfJob {
  JgDev d;
  uint32_t self, seq, step, need;
  uint64_t now;
};
// every node's job in the KERNEL ARGUMENTS (3 KB): read through a pointer into global memory the state machine re-loaded
// every JgDev pointer after every store, copied by value the job went to scratch (344 B per lane: sizeof(JgDev)) - as for
// the slow kernels (jg_follower.h: JgFollowerJobs)
struct JgVoteHalfJobs {
  JgVoteHalfJob j[JG_MAX_REPLICAS];
};
#if JG_BLOCK % 64 == 0  // (the kernels: whole waves; the one-lane host build of the tests calls the per-partition functions above)
// ---- a workgroup over the SET BITS of a bitmap ----------------------------------------------------------------------
// A round's mail names a few percent of the partitions (40 k campaigns at 1 M x 5 and 1 %/round).  With a lane per
// PARTITION nearly every wave of the receiving half held one or two lanes with mail and waited for their dozen dependent
// jg_apply calls with the other sixty idle (733 us per round on the MI355X, the chip latency-bound at 3 % lane
// utilisation); with a lane per bitmap WORD the few waves there are walk their words' bits one after the other.  So: a
// workgroup takes a CHUNK of the bitmap (JG_VOTE_CHUNK words = 4096 partitions; thread k < JG_VOTE_CHUNK brings word k),
// the words and their prefix popcounts go to LDS, and lane i takes the i-th set bit - whole waves at work, the rest of
// the workgroup idle.
#ifndef JG_VOTE_CHUNK
#define JG_VOTE_CHUNK 64u  // (32 until round 6: the receiving half 52.9 -> 45.8 us per round; 16: 0.41 ms per round instead of 0.36; 128: as 64 - profiles/r06/ab_vote_chunk.txt)
#endif
static_assert(JG_BLOCK >= JG_VOTE_CHUNK, "a thread per word of the chunk");
struct JgBitChunk {
  uint64_t w[JG_VOTE_CHUNK];
  uint32_t pre[JG_VOTE_CHUNK + 1];
};
// every thread calls it (threads >= JG_VOTE_CHUNK with word = 0); returns the number of set bits (workgroup-uniform)
__device__ __forceinline__ uint32_t jg_chunk_scan(JgBitChunk& s, uint64_t word) {
  uint32_t total = 0;
  __syncthreads();  // (the last chunk's lanes are done with s)
  const uint32_t ex = jg_block_exclusive_scan(threadIdx.x < JG_VOTE_CHUNK ? (uint32_t)__popcll(word) : 0u, &total);
  if (threadIdx.x < JG_VOTE_CHUNK) s.w[threadIdx.x] = word, s.pre[threadIdx.x] = ex;
  if (threadIdx.x == 0) s.pre[JG_VOTE_CHUNK] = total;
  __syncthreads();
  return total;
}
// the position within the chunk (word * 64 + bit) of its i-th set bit, i < total
__device__ __forceinline__ uint32_t jg_chunk_pick(const JgBitChunk& s, uint32_t i) {
  uint32_t lo = 0, hi = JG_VOTE_CHUNK;  // the word that holds it: the last k with pre[k] <= i
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (s.pre[mid] <= i) lo = mid;
    else hi = mid;
  }
  uint64_t u = s.w[lo];
  for (uint32_t k = i - s.pre[lo]; k; k--) u &= u - 1;
  return lo * 64u + (uint32_t)__ffsll((unsigned long long)u) - 1u;
}

// the receiving half, every node in one launch (blockIdx.y); the bitmap is the addressee's wordmail & ~rowmail (mail that
// came as rows is the row path's).  What a partition costs is very uneven - a voter applies a campaign's R - 1 copies
// (two of them run), a CANDIDATE the R - 1 answers of each of R - 1 voters - and a wave takes as long as its slowest
// lane: the lanes that read answers go first (a second compaction in LDS), so that they share waves with each other.
// (HBM: the bitmaps - G / 4 bytes per node - and, per partition with mail, the senders' control words and its columns.)
__device__ __forceinline__ void jg_vote_half_body(const JgVoteHalfJobs& jobs, uint32_t node, const JgVoteMail& in, const JgVoteMail& out) {
  const JgVoteHalfJob& j = jobs.j[node];
  __shared__ JgBitChunk s;
  __shared__ uint32_t s_g[JG_BLOCK];
  __shared__ uint32_t s_st[JG_VOTE_ST_WORDS][JG_BLOCK];  // the lanes' stretch cursors (jg_vote_half_group): 16 KB
  uint32_t dec = 0;
  const uint32_t n_chunks = (in.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK;
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {  // (block-uniform trip counts throughout)
    const uint32_t w = c * JG_VOTE_CHUNK + threadIdx.x;
    const size_t at = (size_t)j.self * in.words + w;
    const uint32_t total = jg_chunk_scan(s, threadIdx.x < JG_VOTE_CHUNK && w < in.words ? in.wordmail[at] & ~in.rowmail[at] : 0ull);
    for (uint32_t t0 = 0; t0 < total; t0 += JG_BLOCK) {
      const uint32_t n = min(total - t0, (uint32_t)JG_BLOCK);
      uint32_t g = 0, heavy = 0;
      if (threadIdx.x < n) {
        g = c * JG_VOTE_CHUNK * 64u + jg_chunk_pick(s, t0 + threadIdx.x);
        for (uint32_t q = 0; q < in.R; q++) {
          const uint32_t ac = in.rec[jg_vote_at(in, q, g)].a_ctl;
          heavy |= (q != j.self && (ac & 0xffu) && ((ac >> 21) & 7u) == j.self) ? 1u : 0u;
        }
      }
      uint32_t n_heavy = 0;
      const uint32_t ex = jg_block_exclusive_scan(heavy, &n_heavy);
      if (threadIdx.x < n) s_g[heavy ? ex : n_heavy + (threadIdx.x - ex)] = g;
      __syncthreads();
      if (threadIdx.x < n) dec += jg_vote_half_group(j.d, s_g[threadIdx.x], j.self, in, out, j.need, j.now, j.seq, j.step, &s_st[0][threadIdx.x], JG_BLOCK);
      __syncthreads();  // (s_g is the next pass's)
    }
  }
  if (dec) (void)__hip_atomic_fetch_add(&j.d.blk_decisions[blockIdx.x], (uint64_t)dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(JG_BLOCK) void k_vote_half_multi(JgVoteHalfJobs jobs, JgVoteMail in, JgVoteMail out) { jg_vote_half_body(jobs, blockIdx.y, in, out); }
// THE HEAD OF A ROUTED ROUND IN ONE LAUNCH: the receiving half on every node (blockIdx.y < n_vote) and, beside it, the step
// that applies what the last round delivered as ROWS (the other values of blockIdx.y: a job each, jg_apply_runs_body's small
// tiles) - other partitions than the words' (a partition's mail of a round is words or rows, never both), both under the
// delivered step's number, both appending to queues that are unordered by design.  Each of them alone leaves most of the
// chip idle - a few hundred workgroups of dependent loads, one wave per SIMD at work - and one behind the other they were
// 46 + 34 us of a round; side by side on two streams (tried earlier in round 6) the two cross-queue dependencies cost more than
// the overlap gained.
__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_round_head_multi(JgVoteHalfJobs vjobs, uint32_t n_vote, JgVoteMail in, JgVoteMail out,
                                                                        const JgApplyJob* __restrict__ jobs) {
  // (dispatch order: blockIdx.y ascending.  The rows' workgroups go FIRST - there are few of them and each is long, a tile of
  // runs walked lane by lane; the receiving half's many workgroups fill the chip around them)
  const uint32_t n_rows = gridDim.y - n_vote;
  if (blockIdx.y >= n_rows) {
    jg_vote_half_body(vjobs, blockIdx.y - n_rows, in, out);
  } else {
    const JgApplyJob& j = jobs[blockIdx.y];
    jg_apply_runs_body<JG_KINDS_ALL, JG_RUN_TILE_SMALL>(j.d, j.a);
  }
}
// the validation (jg_votes_validate_group) of the partitions a wordmail bit names (any addressee's)
__global__ __launch_bounds__(JG_BLOCK) void k_votes_validate(JgVoteMail m, uint32_t need) {
  __shared__ JgBitChunk s;
  const uint32_t n_chunks = (m.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK;
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const uint32_t w = c * JG_VOTE_CHUNK + threadIdx.x;
    uint64_t u = 0;
    if (threadIdx.x < JG_VOTE_CHUNK && w < m.words)
      for (uint32_t d = 0; d < m.R; d++) u |= m.wordmail[(size_t)d * m.words + w];
    const uint32_t total = jg_chunk_scan(s, u);
    for (uint32_t i = threadIdx.x; i < total; i += JG_BLOCK) jg_votes_validate_group(m, c * JG_VOTE_CHUNK * 64u + jg_chunk_pick(s, i), need);
  }
}
// a round's mail cleared for its next use - where it was written: a control word is only ever written together with a
// wordmail bit of its partition (the census's first copy, the receiving half's answer), so the union of the addressees'
// wordmail words says which partitions' control words are dirty (R x G / 8 bytes read instead of 8 x R x G bytes written
// per round: 0.6 MB against 40 MB at 1 M x 5); then the bitmaps themselves (the thread that brought a word clears it: the
// lanes work from the copy in LDS).
// (`a`, `b`: two more word ranges that go back to zero with the round's mail - the transport's tallies and its bucket
// counters, k_route_clear's job: one launch less per round; `cp_*`: the round's job tables on their way from the host's pinned
// staging to the device, k_copy_words' job: another one)
__device__ __forceinline__ void jg_votes_clear_mail(const JgVoteMail& m) {
  __shared__ JgBitChunk s;
  const uint32_t n_chunks = (m.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK;
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const uint32_t w = c * JG_VOTE_CHUNK + threadIdx.x;
    uint64_t u = 0;
    if (threadIdx.x < JG_VOTE_CHUNK && w < m.words)
      for (uint32_t d = 0; d < m.R; d++) {
        u |= m.wordmail[(size_t)d * m.words + w];
        m.wordmail[(size_t)d * m.words + w] = 0, m.rowmail[(size_t)d * m.words + w] = 0;
      }
    const uint32_t total = jg_chunk_scan(s, u);
    for (uint32_t i = threadIdx.x; i < total; i += JG_BLOCK) {
      const uint32_t g = c * JG_VOTE_CHUNK * 64u + jg_chunk_pick(s, i);
      for (uint32_t q = 0; q < m.R; q++) *(uint64_t*)&m.rec[jg_vote_at(m, q, g)].q_ctl = 0;  // (q_ctl | a_ctl: one 8-byte store)
    }
  }
}
// (m.words = 0: no mail to clear - the round's mail was cleared beside the injected rows' step of the round before, k_apply_rows_clear_multi)
__global__ __launch_bounds__(JG_BLOCK) void k_votes_clear(JgVoteMail m, uint32_t* __restrict__ a, uint32_t na, uint32_t* __restrict__ b, uint32_t nb,
                                                          uint64_t* __restrict__ cp_dst = nullptr, const uint64_t* __restrict__ cp_src = nullptr, uint32_t cp_n = 0) {
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < cp_n; i += gridDim.x * JG_BLOCK) cp_dst[i] = cp_src[i];
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < na + nb; i += gridDim.x * JG_BLOCK) {
    if (i < na) a[i] = 0;
    else b[i - na] = 0;
  }
  jg_votes_clear_mail(m);
}
// The injected rows' step of a routed round (blockIdx.y < n_jobs: a node's batch each, jg_apply_rows_body) and, beside it, the
// clearing of the mail the round's receiving half has just read - the NEXT round's to fill (the last value of blockIdx.y): the
// step is a hundred workgroups of dependent loads, the clearing ten microseconds of a launch of its own at the head of every
// round until the end of round 6.
__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_apply_rows_clear_multi(const JgApplyJob* __restrict__ jobs, uint32_t n_jobs, JgVoteMail m) {
  if (blockIdx.y < n_jobs) {
    const JgApplyJob& j = jobs[blockIdx.y];
    jg_apply_rows_body(j.d, j.a);
  } else {
    jg_votes_clear_mail(m);
  }
}
#endif
