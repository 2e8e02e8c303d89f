/*
 * josefine_gpu.h — C ABI of the MI355X batched Chained-Raft engine.
 *
 * This is the drop-in boundary for the hot path of tychedelia/josefine's
 * `src/raft` state machine.  One engine hosts N independent
 * Raft node instances ("groups", index g) and applies batches of
 * `Command`s to them with exactly the per-instance semantics of the reference's
 *
 *     trait Apply { fn apply(self, cmd: Command) -> Result<RaftHandle>; }
 *                                               (src/raft/mod.rs:483-489)
 *
 * Plain pointers and sizes only: this header is what a Rust `extern "C"`
 * block / bindgen would bind (see INTEGRATION.md for the adapter that
 * implements `Apply` on top of it).  All paths cited below are relative to the
 * reference checkout.
 *
 * Conventions
 *   - every entry point returns 0 (JG_OK) or a negative JG_E* status;
 *     `jg_last_error()` returns a thread-local description of the last failure;
 *   - reference panics / `Err` returns never abort a batch: they are recorded as
 *     a sticky per-group fault code (`JG_FIELD_FAULT`), after which the group
 *     ignores commands until `JG_CMD_RESTART` (the process would be dead);
 *   - BlockId is the reference's 8-byte big-endian id (src/raft/chain.rs:29-36,
 *     63-67) carried as a native uint64_t (same ordering);
 *   - one engine is externally synchronised (one caller thread at a time),
 *     exactly like the by-value `RaftHandle` (src/raft/mod.rs:471-479).
 */
#ifndef JOSEFINE_GPU_H
#define JOSEFINE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JG_ABI_VERSION 9u /* v9: JG_NODE_KEEP (two node steps in flight: jg_node_outbox_view serves the oldest); v8: jg_dense_cluster_round_routed delivers a partition's mail in the order (phase, emission index, sender) instead of
                             (sender, step, emission index) - same entry points, same layouts, different (legal) network schedule; v5: jg_step_node = arrival-order Apply (fsm_tx: Apply / Notify / Apply per partition), JG_CLUSTER_ANY_LEADER,
                             jg_dense_cluster_withdraw_appends, JG_COL_UNCHECKED, JG_COL_UPLOAD_NOW; v6: jg_dense_cluster_set_option, jg_dense_cluster_offer_appends, JG_CMD_RECREATE;
                             v7: the node step's bus formats - JG_COL_PACKED_KIND, JG_COL_ID32, JG_NODE_COMMON_AE (jg_node_outbox.aec), JG_NODE_FSM_FUSED (JG_FSM_LEADER_STEP) */
#define JG_MAX_REPLICAS 8u  /* R <= 8: vote / progress-state masks are one byte            */
#define JG_CHAIN_WINDOW 8u  /* chain segments (gaps / forks) per group besides the main run */
#define JG_MAX_INFLIGHT 5u  /* src/raft/progress.rs:117                                    */
#define JG_FOREIGN_VOTERS 8u /* distinct voters outside the membership an election remembers */
#define JG_NO_ACK UINT64_MAX /* dense ack column: "no AppendResponse from this replica"    */
#define JG_MAX_DEVICES 16u  /* shards (devices) behind one engine handle                      */
#define JG_MAX_DENSE_APPENDS (1u << 20) /* own slot of a dense ack block: appends per group per tick */

/* ---- status codes -------------------------------------------------------- */
enum {
  JG_OK = 0,
  JG_EINVAL = -1,   /* bad argument / config (cf. RaftConfig::validate, src/raft/config.rs:60-84) */
  JG_ENOMEM = -2,
  JG_EDEVICE = -3,  /* HIP runtime error; no CPU fallback exists                    */
  JG_ECAPACITY = -4 /* caller's output buffer too small; nothing was consumed       */
};

/* ---- roles: RaftHandle variants, src/raft/mod.rs:417-425 ------------------ */
enum { JG_ROLE_FOLLOWER = 0, JG_ROLE_CANDIDATE = 1, JG_ROLE_LEADER = 2 };

/* ---- Command kinds: enum Command, src/raft/mod.rs:160-227 ------------------
 * Column use per kind (unused columns are ignored):
 *   kind              from          term   id            aux        flag
 *   TICK              -             -      -             -          -
 *   PROPOSE           -             -      -             -          -
 *   VOTE_REQUEST      candidate_id  term   head          last_term  -
 *   VOTE_RESPONSE     from          term   -             -          granted
 *   APPEND_ENTRIES    leader_id     term   first block*  n_blocks   -
 *   APPEND_RESPONSE   node_id       term   head          -          success
 *   HEARTBEAT         leader_id     term   commit        -          -
 *   HEARTBEAT_RESPONSE (sender**)   -      commit        -          has_committed
 *   TIMEOUT           -             -      -             -          -
 *   NOOP              -             -      -             -          -
 *   CLIENT_REQUEST    -             -      request token -          -
 *   CLIENT_RESPONSE   -             -      request token -          -
 *   RESTART (engine)  -             -      -             -          -
 *   RECREATE (engine) -             -      -             -          -
 * (*) index of the first of `aux` consecutive entries in the batch's
 *     blk_id/blk_next side arrays (Vec<Block>, payload stays on the host).
 * (**) the reference's handler does not look at the sender (leader.rs:222-231) and neither does
 *     jg_step; jg_step_node needs Message.from (rpc.rs:17-27) to put the response into the
 *     sender's mailbox slot (0 or a non-member: the row takes the general path).
 * RESTART is not a reference Command: it restates process restart, i.e.
 * `Raft::<Follower>::new` + `Chain::new` on the persisted tree
 * (src/raft/follower.rs:68-95, src/raft/chain.rs:117-137), and clears the fault.
 */
enum {
  JG_CMD_TICK = 0,
  JG_CMD_PROPOSE = 1,
  JG_CMD_VOTE_REQUEST = 2,
  JG_CMD_VOTE_RESPONSE = 3,
  JG_CMD_APPEND_ENTRIES = 4,
  JG_CMD_APPEND_RESPONSE = 5,
  JG_CMD_HEARTBEAT = 6,
  JG_CMD_HEARTBEAT_RESPONSE = 7,
  JG_CMD_TIMEOUT = 8,
  JG_CMD_NOOP = 9,
  JG_CMD_CLIENT_REQUEST = 10,
  JG_CMD_CLIENT_RESPONSE = 11,
  JG_CMD_RESTART = 12,
  JG_CMD_RECREATE = 13,
  JG_CMD__COUNT = 14
};
/* The two engine commands (no reference Command): JG_CMD_RESTART = the replica's process restarts on its persisted tree
 * (Raft::new + Chain::new on the sled directory it left: follower.rs:68-95, chain.rs:117-137); JG_CMD_RECREATE = on an
 * EMPTY directory - the replica of a partition that was re-created: genesis only, no "commit" key (chain.rs:117-153), and
 * the host's block store for the group starts over with it.  Both clear a fault; neither looks at the other columns. */

/* ---- Address, src/raft/rpc.rs:5-14 ---------------------------------------- */
enum { JG_TO_PEERS = 0, JG_TO_PEER = 1, JG_TO_LOCAL = 2, JG_TO_CLIENT = 3, JG_TO_QUEUE = 4 };

/* ---- client request queue rows ----------------------------------------------
 * Follower / Candidate queue client requests they cannot forward yet
 * (src/raft/follower.rs:22,258-270; candidate.rs:20,190-193).  The request
 * payloads stay with the host adapter; the engine keeps only the queue length
 * and tells the adapter what to do with its mirror through CLIENT_REQUEST rows:
 *   to_kind JG_TO_QUEUE, flag 0              : push request `id` onto the group's queue
 *   to_kind JG_TO_PEER,  flag JG_QUEUE_FLUSH : send the `aux` queued requests, in order, to
 *                                              to_id, then clear the queue (follower.rs:190-197)
 *   to_kind JG_TO_QUEUE, flag JG_QUEUE_DROP  : the `aux` queued requests are dropped by a
 *                                              role change (follower.rs:295, candidate.rs:216-238)
 *   to_kind JG_TO_PEER,  flag 0              : forward request `id` now (follower.rs:262-265) */
enum { JG_QUEUE_FLUSH = 1, JG_QUEUE_DROP = 2 };

/* ---- fault codes ------------------------------------------------------------
 * 1..63: the reference would panic or return Err at the cited line.
 * 128.. : engine-domain limits (inputs outside what the device layout can
 *         represent); never silently wrong, always loud. */
enum {
  JG_FAULT_NONE = 0,
  JG_FAULT_LEADER_TERM_UNIMPLEMENTED = 1, /* leader.rs:33-35 via leader.rs:200-208, mod.rs:360-365 */
  JG_FAULT_APPEND_ID_NOT_ABOVE_HEAD = 2,  /* chain.rs:163 assert!(id > self.head)                  */
  JG_FAULT_PROGRESS_UNKNOWN_NODE = 3,     /* progress.rs:43 expect("the node does not exist")      */
  JG_FAULT_COMMIT_MISSING_BLOCK = 4,      /* chain.rs:197-202 panic!("")                           */
  JG_FAULT_EXTEND_MISSING_PARENT = 5,     /* chain.rs:180-185 Err via follower.rs:159              */
  JG_FAULT_FOLLOWER_STALE_LEADER = 6,     /* follower.rs:147-154 assert!                           */
  JG_FAULT_CANDIDATE_TICK_ELECTED = 7,    /* candidate.rs:64 panic!("this should never happen")    */
  JG_FAULT_RANGE_HIT_COMMIT_KEY = 8,      /* chain.rs:198 + 219-226 via leader.rs:135,152-157 (Q9) */
  JG_FAULT_ENGINE_WINDOW_OVERFLOW = 128,  /* chain needs > JG_CHAIN_WINDOW segments (gaps / forks) */
  JG_FAULT_ENGINE_FOREIGN_VOTER = 129,    /* a 9th distinct voter outside the membership in one election
                                             (the first JG_FOREIGN_VOTERS are counted, election.rs:33-35) */
  JG_FAULT_ENGINE_DENSE_NONLEADER = 131,  /* dense tick asked a non-leader group to append         */
  JG_FAULT_ENGINE_DENSE_APPENDS = 132,    /* own slot of a dense ack block >= JG_MAX_DENSE_APPENDS */
  JG_FAULT_ENGINE_MAILBOX_RANGE = 133     /* a block id >= 2^56 - 1 would have to go into a mailbox word */
};

/* ---- engine configuration ---------------------------------------------------
 * Restates the parts of RaftConfig that feed L1 (src/raft/config.rs:14-41,
 * defaults 87-111) for N instances at once.  `node_ids[r]` is the NodeId of
 * replica slot r; an instance's own slot is `self_slot[g]` (default 0) and its
 * `config.nodes` are the other slots in ascending slot order. */
typedef struct jg_config {
  uint32_t abi_version;              /* JG_ABI_VERSION                                        */
  uint32_t n_groups;                 /* G                                                     */
  uint32_t n_replicas;               /* R = config.nodes.len()+1, 1..JG_MAX_REPLICAS          */
  uint32_t node_ids[JG_MAX_REPLICAS];/* nonzero, distinct (config.rs:64-66)                   */
  int32_t device_id;                 /* HIP device ordinal                                    */
  uint32_t heartbeat_timeout_ms;     /* config.rs:104 (100)                                   */
  uint32_t election_timeout_min_ms;  /* mod.rs:318 (500)                                      */
  uint32_t election_timeout_max_ms;  /* mod.rs:319 (1000)                                     */
  uint64_t seed;                     /* replaces thread_rng (follower.rs:105); see DESIGN.md  */
  uint64_t group_base;               /* global id of local group 0 (sharding; keys the RNG)   */
  uint32_t flags;                    /* JG_CFG_*                                              */
  uint32_t reserved;
  /* Multi-device engine (SURVEY.md §8(b),(e)): n_devices >= 1 shards the G groups over
   * device_ids[0..n_devices) by contiguous ownership — shard d owns local groups
   * [d*S, min((d+1)*S, G)) with S = ceil(G / n_devices) (trailing shards that would be empty are
   * not created) — behind this ONE handle: the reference has one caller that owns the handle
   * (event_loop, src/raft/server.rs:103-165).  A device may be listed more than once (several
   * shards on one GPU).  n_devices == 0: one shard on `device_id`. */
  uint32_t n_devices;
  int32_t device_ids[JG_MAX_DEVICES];
} jg_config;

enum {
  /* keep the "commit" key out of the block keyspace, i.e. do NOT reproduce Q9
   * (SURVEY.md §7.3): unbounded ranges then simply end at the last block. */
  JG_CFG_SEPARATE_COMMIT_KEY = 1u,
  /* jg_step_node's row passes UNTILED: one thread per row scatters into the G-sized columns (three random 4-8-byte
   * accesses per row and pass) instead of binning the rows by tile of 256 partitions and applying a tile's rows to its
   * columns in LDS.  The results are the same bit for bit; the flat passes are the statement the tiled ones are held to
   * (tests/test_node_step.py) and an A/B switch - nothing to set in production. */
  JG_CFG_FLAT_ROW_PASSES = 2u
};

/* ---- SoA command batch (host memory), SURVEY.md §8(a) a18 -------------------- */
typedef struct jg_cmd_batch {
  size_t n;               /* number of command rows                                       */
  const uint8_t* kind;    /* [n] JG_CMD_*                                                 */
  const uint32_t* group;  /* [n] local group index; rows of one group apply in row order  */
  const uint32_t* from;   /* [n] NodeId                                                   */
  const uint64_t* term;   /* [n]                                                          */
  const uint64_t* id;     /* [n] BlockId / token / side-array index                       */
  const uint64_t* aux;    /* [n]                                                          */
  const uint8_t* flag;    /* [n]                                                          */
  size_t n_blocks;        /* entries in the side arrays                                   */
  const uint64_t* blk_id; /* [n_blocks] Block.id   (chain.rs:86-91)                       */
  const uint64_t* blk_next;/*[n_blocks] Block.next                                        */
} jg_cmd_batch;

/* ---- outbound Message rows: what the reference pushes on rpc_tx -----------
 * (src/raft/mod.rs:337-340,390-400; src/raft/rpc.rs:17-27).  `kind` is the
 * Command kind; columns as in the table above, with `from` = sender NodeId.
 * APPEND_ENTRIES rows (leader.rs:124-174) carry id = range start key
 * (progress.head) and aux = number of blocks: the payload is the next `aux`
 * stored blocks after skipping the first item of `range(id..)`. */
typedef struct jg_msg_row {
  uint32_t group;
  uint8_t kind;
  uint8_t to_kind; /* JG_TO_* */
  uint8_t flag;
  uint8_t pad;
  uint32_t to_id;  /* NodeId when to_kind == JG_TO_PEER */
  uint32_t from;
  uint64_t term;
  uint64_t id;
  uint64_t aux;
} jg_msg_row;

/* ---- FSM instruction rows: what the reference pushes on fsm_tx -------------
 * (src/raft/fsm.rs:20-29).  Apply instructions are run-length encoded as key
 * ranges; the host expands them against its block store in key order:
 *   JG_FSM_APPLY_LEADER   : range(a..=b).skip(1)   (leader.rs:93)
 *   JG_FSM_APPLY_FOLLOWER : range(a..b)            (follower.rs:204, half-open)
 *   JG_FSM_NOTIFY         : Notify{block_id=a, id=b} (leader.rs:184-188)
 *   JG_FSM_LEADER_STEP    : (jg_step_node with JG_NODE_FSM_FUSED only) everything a leader partition pushed in
 *                           one step as ONE row: a = the appended block's id, b = the ClientRequest's token,
 *                           pad[0..2] = a - c0, a - c1, a - c2 (each <= 255) standing for, in this order,
 *                             JG_FSM_APPLY_LEADER {c0, c1} if c1 != c0;  JG_FSM_NOTIFY {a, b};
 *                             JG_FSM_APPLY_LEADER {c1, c2} if c2 != c1 */
enum { JG_FSM_APPLY_LEADER = 0, JG_FSM_APPLY_FOLLOWER = 1, JG_FSM_NOTIFY = 2, JG_FSM_LEADER_STEP = 3 };
typedef struct jg_fsm_row {
  uint32_t group;
  uint8_t kind;
  uint8_t pad[3];
  uint64_t a;
  uint64_t b;
} jg_fsm_row;

typedef struct jg_fault_row {
  uint32_t group;
  uint32_t code; /* JG_FAULT_* */
} jg_fault_row;

/* ---- state columns readable through jg_read_state ------------------------- */
enum {
  JG_FIELD_TERM = 0,        /* u64  State.current_term          mod.rs:277            */
  JG_FIELD_VOTED_FOR = 1,   /* u32  State.voted_for or 0        mod.rs:279            */
  JG_FIELD_HAS_VOTED = 2,   /* u8   voted_for.is_some()                                */
  JG_FIELD_ROLE = 3,        /* u8   JG_ROLE_*                   mod.rs:417-425        */
  JG_FIELD_COMMIT = 4,      /* u64  Chain.commit                chain.rs:102          */
  JG_FIELD_HEAD = 5,        /* u64  Chain.head                  chain.rs:103          */
  JG_FIELD_ID_GEN = 6,      /* u64  Chain.id_gen                chain.rs:101          */
  JG_FIELD_MATCH = 7,       /* u64  Progress.head of `replica`  progress.rs:124       */
  JG_FIELD_REPL_STATE = 8,  /* u8   bit r = Replicate (else Probe) progress.rs:62-66  */
  JG_FIELD_VOTE_SEEN = 9,   /* u8   bit r = votes.contains(r)   election.rs:8         */
  JG_FIELD_VOTE_GRANTED = 10,/*u8   bit r = votes[r] == true                           */
  JG_FIELD_FAULT = 11,      /* u8   JG_FAULT_*                                         */
  JG_FIELD_LEADER_ID = 12,  /* u32  Follower.leader_id or 0     follower.rs:20        */
  JG_FIELD_HAS_LEADER = 13, /* u8                                                      */
  JG_FIELD_ELECTION_TIME = 14,   /* u64 ms, State.election_time       mod.rs:281      */
  JG_FIELD_ELECTION_TIMEOUT = 15,/* u32 ms, State.election_timeout    mod.rs:283      */
  JG_FIELD_HEARTBEAT_TIME = 16,  /* u64 ms, Leader.heartbeat_time     leader.rs:27    */
  JG_FIELD_QUEUED_REQS = 17,     /* u32 queued_reqs.len()  follower.rs:22, candidate.rs:20 */
  JG_FIELD_SELF_SLOT = 18,  /* u8                                                      */
  JG_FIELD__COUNT = 19
};

typedef struct jg_engine jg_engine;

/* RaftHandle::new for every group (src/raft/mod.rs:428-435 -> follower.rs:68-95):
 * Follower, State::default(), fresh Chain (genesis block 0), election timer armed
 * at now = 0.  Fails with JG_EDEVICE when no gfx950 device / HIP runtime is usable. */
int jg_engine_create(const jg_config* cfg, jg_engine** out);
void jg_engine_destroy(jg_engine* e);

/* ---- shards of a multi-device engine ------------------------------------------------------------
 * Everything that takes HOST memory works on the parent handle exactly as on a single-device
 * engine: jg_submit buckets the rows by owner (stable: a group's rows keep their order), jg_step
 * steps every shard (one host thread and one HIP stream per shard), the drains merge the shards'
 * rows back into the single-engine order (per step: groups ascending; steps in order),
 * jg_read_state / jg_get_counters / jg_drain_faults / jg_sync / jg_set_self_slots /
 * jg_step_dense_acks (host [R][G] block) split or concatenate per shard.  Results are
 * bit-identical to one engine over the same G groups (tests/test_multi_device.py).
 * Everything that takes DEVICE pointers is per device by nature: the caller addresses a shard's
 * own engine handle (jg_get_shard; an ordinary single-device engine over groups
 * [group_lo, group_lo + n_groups), local group index = global - group_lo) for
 * jg_step_dense_leader / _follower, jg_step_device_rows, jg_device_*, jg_synth_fill_acks_device,
 * jg_calibrate_stream; jg_step_dense_acks_shards launches the ack tick on every shard at once.
 * Shard handles are owned by the parent: never destroy them, and do not drain them directly. */
typedef struct jg_shard_info {
  jg_engine* engine;  /* the shard's own single-device engine                                  */
  int32_t device_id;
  uint32_t group_lo;  /* first group (index within the parent) this shard owns                 */
  uint32_t n_groups;
  uint32_t reserved;
} jg_shard_info;
uint32_t jg_shard_count(const jg_engine* e); /* 1 for a single-device engine (which is its own shard) */
int jg_get_shard(jg_engine* e, uint32_t shard, jg_shard_info* out);

/* Per-group own replica slot (default: all 0).  Only legal before the first step. */
int jg_set_self_slots(jg_engine* e, const uint8_t* slots /* [G] host */);

/* Queue a batch of commands (host memory, borrowed for the call).  Rows may be in
 * any group order; rows of the same group are applied in row order, after rows
 * queued by earlier jg_submit calls. */
int jg_submit(jg_engine* e, const jg_cmd_batch* batch);

/* Zero-copy jg_submit for a caller that produces rows itself (a transport's receive task decoding
 * straight into the engine's pinned columns, src/raft/tcp.rs:139-170 -> server.rs:126-137): room for n
 * more rows and n_blocks side-array entries at the tail of the pending batch; the caller fills kind,
 * group, id and whichever optional columns it needs in place (an AppendEntries row's id indexes THIS
 * reservation's side-array entries) and commits how many it wrote, and which optional columns
 * (JG_COL_*): a column not named is all zeros for these rows and costs nothing - jg_step_node does not
 * even upload a column that no submit of the step provided.  jg_submit_commit checks what jg_submit
 * checks.  The pointers are good until the next jg_submit* or step call.  Single-device engines (or a
 * shard's own handle). */
typedef struct jg_cmd_cols {
  uint8_t* kind;
  uint32_t* group;
  uint32_t* from;
  uint64_t* term;
  uint64_t* id;
  uint64_t* aux;
  uint8_t* flag;
  uint64_t* blk_id;
  uint64_t* blk_next;
} jg_cmd_cols;
enum { JG_COL_FROM = 1u, JG_COL_TERM = 2u, JG_COL_AUX = 4u, JG_COL_FLAG = 8u,
       /* no validation pass over the rows on the host (it costs 2.5 ms per 9 M rows): jg_step_node's classification
        * checks group and kind on the device - a row out of range is not applied and the next synchronising call
        * returns JG_EINVAL.  Only jg_step_node takes such a batch (jg_step refuses it). */
       JG_COL_UNCHECKED = 16u,
       /* these rows are the whole batch of the next jg_step_node (nothing was committed before them since the last
        * step, nothing follows): their upload starts NOW, on a copy stream of its own - while the previous step's
        * kernels run and its outputs travel the other way - instead of at the head of the step.  A batch that turns
        * out not to be the step's whole input is simply uploaded again by the step. */
       JG_COL_UPLOAD_NOW = 32u,
       /* the kind column of these rows holds `kind | sender << 4 | flag << 7`: sender = the member slot (index into
        * jg_config.node_ids, 3 bits) of Message.from for the kinds that carry one (VoteRequest ... HeartbeatResponse;
        * a slot >= n_replicas reads as NodeId 0, the other kinds read 0 whatever the bits), flag as the flag column -
        * neither a `from` nor a `flag` column exists for the step (naming JG_COL_FROM / JG_COL_FLAG with it is an
        * error): an AppendResponse row is 13 bytes on the bus, not 18.  Every commit of a step must agree on it, it
        * needs JG_COL_UNCHECKED (the device decodes and checks the byte), and jg_submit cannot add rows to such a step. */
       JG_COL_PACKED_KIND = 64u,
       /* the id column of these rows holds 32-bit values: the caller writes n uint32_t at (uint32_t*)cols.id (block ids,
        * commit indices, request tokens, side-array indices - zero-extended on the device): 4 bytes per row on the bus
        * instead of 8.  Only as the step's ONE commit (no rows before it, none after it, no jg_submit in the step); needs
        * JG_COL_UNCHECKED. */
       JG_COL_ID32 = 128u };
int jg_submit_reserve(jg_engine* e, size_t n, size_t n_blocks, jg_cmd_cols* cols);
int jg_submit_commit(jg_engine* e, size_t n, size_t n_blocks, uint32_t optional_columns);

/* Apply everything queued since the last step: for each group, `apply(cmd)` in
 * stream order (src/raft/mod.rs:471-479).  `now_ms` is the logical clock that
 * replaces Instant::now() (mod.rs:352-357, follower.rs:110-113, leader.rs:78-84).
 * Outputs are appended to the message / fsm queues.  Asynchronous on the
 * engine's stream; drains and reads synchronise. */
int jg_step(jg_engine* e, uint64_t now_ms);

/* The same step for a batch that already lives in device memory: every pointer of
 * `dev_batch` is a device pointer (all seven columns required; the block side
 * arrays only if AppendEntries rows are present) and the rows MUST be sorted by
 * group, rows of one group in stream order — the form k_apply_rows consumes, so no
 * host pass is needed.  Unsorted rows or an out-of-range group are reported as
 * JG_EINVAL by the next synchronising call. */
int jg_step_device_rows(jg_engine* e, const jg_cmd_batch* dev_batch, uint64_t now_ms);

/* Dense steady-state leader tick (the HBM-roofline path).
 * `acks` is an [R][G] column-major array (replica-major: acks[r*G+g]):
 *   r != self_slot[g]: head of an AppendResponse{node_id: node_ids[r], head} or JG_NO_ACK;
 *   r == self_slot[g]: number of ClientRequests to append this tick, < JG_MAX_DENSE_APPENDS
                        (a larger value, JG_NO_ACK included, raises JG_FAULT_ENGINE_DENSE_APPENDS
                        on that group and applies nothing of its tick).
 * Per group, in this order: the appends (leader.rs:177-197, each with its
 * self-ack), then the acks in ascending slot order (leader.rs:211-219 ->
 * progress.rs:42-60 -> leader.rs:87-99).  Equivalent to submitting those
 * commands through jg_submit/jg_step, except that the FSM instructions are not
 * queued: they are the per-group head / commit deltas (Notify for ids
 * (head_before, head_after], Apply for keys (commit_before, commit_after]), which
 * the caller reads back with jg_read_state.  Non-leader groups ignore acks exactly as the
 * reference does (follower.rs:62, candidate.rs:194); asking one to append is a
 * precondition violation of the dense path (client proxying needs the message
 * queue of jg_step) and raises JG_FAULT_ENGINE_DENSE_NONLEADER. */
int jg_step_dense_acks(jg_engine* e, const uint64_t* acks_host);
int jg_step_dense_acks_device(jg_engine* e, const uint64_t* acks_dev);
/* `n_ticks` consecutive dense ticks in one launch: tick t reads the [R][G] block at
 * acks_dev + t*R*G.  Identical in effect to n_ticks calls of jg_step_dense_acks_device;
 * the groups' state is read and written once per launch instead of once per tick. */
int jg_step_dense_acks_device_n(jg_engine* e, const uint64_t* acks_dev, uint32_t n_ticks);
/* The same for every shard of a multi-device engine: acks_dev[d] is shard d's own [n_ticks][R][G_d]
 * block in the memory of its device (G_d = its n_groups).  One launch per shard, issued by the
 * shard's host thread on the shard's stream; returns when all are enqueued. */
int jg_step_dense_acks_shards(jg_engine* e, const uint64_t* const* acks_dev, uint32_t n_ticks);

/* ---- dense node tick: the steady-state traffic of a cluster in column form ----------------
 * What a leader sends its followers on a Tick (leader.rs:234-245: heartbeat() if due, then
 * replicate()) and what they answer (follower.rs:130-217), as device-resident SoA mailboxes, so
 * that a whole protocol round of millions of groups never leaves HBM: one engine's outbox
 * columns are another engine's inbox columns (same device: the same pointers).
 *
 * Mailbox vocabulary (everything else a step emits is queued as ordinary jg_msg_row rows,
 * drained with jg_drain_messages, in per-group emission order).  One 8-byte word per message where
 * the message has one addressee, so that a tick touches as few address streams as possible (the node
 * tick is bound by the number of its memory instructions, not by their bytes: profiles/README.md):
 *   Heartbeat{term, commit, leader_id}      -> beat[g] = {term, commit}
 *   AppendEntries{term, leader_id, blocks}  -> beat[g].term, ae[r][g] = JG_AE(from, n): the blocks are
 *       ids from+1 .. from+n, each with next = id-1 — expressible exactly when the leader's chain is
 *       a run (the id set is [0, top] and every block's parent its predecessor: what append builds -
 *       and what a restarted replica re-opens, its head at its commit index possibly BELOW the top,
 *       chain.rs:117-137; replicate() ranges over the stored keys, leader.rs:135,152-157); a leader
 *       whose chain is not sends all messages of its Tick as rows instead
 *   AppendResponse{node_id, head}           -> answer[g], bits 63..8 (one word of the leader's inbox block)
 *   HeartbeatResponse{commit, has_committed}-> answer[g], bits 7..0 = has_committed; commit -> hb_commit[g]
 * Block ids in mailbox words are 56 bits wide: a group whose head reaches JG_MAILBOX_NONE raises
 * JG_FAULT_ENGINE_MAILBOX_RANGE instead of emitting (never silently wrong; sequential ids get there
 * after 7 x 10^16 appends).
 * FSM instructions are not queued by dense steps: they are the per-group commit / head deltas
 * (follower: Apply for keys [commit_before, commit_after), follower.rs:204; leader: as
 * jg_step_dense_acks). */
#define JG_AE_NONE 0xFFu /* low byte of an ae word: no AppendEntries for this slot / group this tick */
#define JG_HB_NONE 0xFFu /* low byte of an answer word: no HeartbeatResponse                        */
#define JG_MAILBOX_NONE 0x00FFFFFFFFFFFFFFull /* bits 63..8 of an answer word: no AppendResponse    */
/* (the all-ones word JG_NO_ACK is therefore "nothing at all" in both kinds of word) */
#define JG_ANSWER(ack_head, hb_code) (((uint64_t)(ack_head) << 8) | (uint64_t)(hb_code))
#define JG_AE(from, n) (((uint64_t)(from) << 8) | (uint64_t)(n))

typedef struct jg_leader_beat { /* what every follower of the group reads of the leader's Tick */
  uint64_t term;      /* current_term of this tick's messages                              */
  uint64_t hb_commit; /* Heartbeat.commit, or JG_NO_ACK: no heartbeat (not due / no leader) */
} jg_leader_beat;

typedef struct jg_leader_inbox { /* device pointers; answers == NULL: nothing came in */
  const uint64_t* answers;   /* [R][G] JG_ANSWER(AppendResponse.head or JG_MAILBOX_NONE, has_committed or
                                JG_HB_NONE) of slot r; own slot: JG_ANSWER(#ClientRequests to append, JG_HB_NONE),
                                the count as in jg_step_dense_acks                                          */
  const uint64_t* hbr_commit;/* [R][G] HeartbeatResponse.commit (read only where has_committed == 0)     */
} jg_leader_inbox;

typedef struct jg_leader_outbox { /* device pointers, all required */
  jg_leader_beat* beat; /* [G]                                                                        */
  uint64_t* ae;         /* [R][G] JG_AE(range start key = progress head of slot r, number of blocks
                           0..JG_MAX_INFLIGHT) (leader.rs:135,152), or JG_NO_ACK: nothing for slot r.
                           Row [own slot] is nobody's mail: while every group of the engine has the same
                           own slot (the default; jg_set_self_slots with one value) the kernel does NOT
                           write it - fill it with JG_NO_ACK once if something walks all R rows (the
                           engine's own blocks - jg_dense_cluster, jg_step_node - are filled that way) */
} jg_leader_outbox;

typedef struct jg_follower_inbox { /* device pointers */
  const uint32_t* leader;   /* [G] sender NodeId per group, or NULL: `leader_id` for every group   */
  uint32_t leader_id;
  uint32_t reserved;
  const jg_leader_beat* beat;/* [G] the leader's outbox beat                                        */
  const uint64_t* ae;       /* [G] this node's row of the leader's outbox ae                       */
} jg_follower_inbox;

typedef struct jg_follower_outbox { /* device pointers, all required */
  uint64_t* answer;    /* [G] JG_ANSWER(AppendResponse.head or JG_MAILBOX_NONE, has_committed or JG_HB_NONE):
                          this node's row of the leader's inbox answers                                 */
  uint64_t* hb_commit; /* [G] HeartbeatResponse.commit (written where a HeartbeatResponse is)          */
} jg_follower_outbox;

/* Leader half of a node tick.  Per group that is a healthy leader, in this order:
 *   1. the HeartbeatResponses of `in`, ascending slot (leader.rs:222-231: replicate() again if
 *      !has_committed && commit > 0 — those extra AppendEntries are queued as rows);
 *   2. the appends and AppendResponses of in->answers exactly as jg_step_dense_acks;
 *   3. if `out` != NULL: Command::Tick (leader.rs:234-245) into the outbox columns.
 * Groups that are not leaders ignore 1-2 as the reference does and are not ticked here
 * (jg_step_dense_follower ticks them): their outbox entries are "none". */
int jg_step_dense_leader(jg_engine* e, uint64_t now_ms, const jg_leader_inbox* in, const jg_leader_outbox* out);

/* Follower half of a node tick.  Per group that is not a leader, in this order:
 *   1. Heartbeat{beat.term, beat.hb_commit, leader}    if beat[g].hb_commit != JG_NO_ACK (follower.rs:178-217)
 *   2. AppendEntries{beat.term, leader, blocks}        if ae[g] != JG_NO_ACK             (follower.rs:130-176)
 *   3. Command::Tick                                   if tick != 0     (follower.rs:121-128, candidate.rs:48-68)
 * Leaders apply 1-2 as the reference does (leader.rs:200-208,263) and are not ticked here: the leader half ticks
 * whoever is a leader when the round begins - also one that steps down in 1-2 (one Tick per group and round).
 * Equivalent to submitting those commands through jg_submit/jg_step, except that
 * AppendResponse / HeartbeatResponse go to the outbox columns and no FSM rows are queued. */
int jg_step_dense_follower(jg_engine* e, uint64_t now_ms, const jg_follower_inbox* in,
                           const jg_follower_outbox* out, int tick);

/* ---- a node's whole tick from host rows: the dense kernels behind the Apply surface ---------------
 * What server::event_loop (src/raft/server.rs:103-165) does between two ticks of its interval, for
 * every partition this node hosts at once: apply whatever tcp_rx / client_rx delivered, one command at a
 * time IN THE ORDER IT ARRIVED (jg_submit: the rows of a partition in their stream order, partitions
 * interleaved in any way, no sorting on the host), then Command::Tick.  The result - state, faults, and per
 * partition the sequence of everything pushed on fsm_tx and rpc_tx - is that of jg_submit + jg_step over the
 * same rows followed by a Tick row per partition: Apply::apply in arrival order (mod.rs:471-479).  Nothing is
 * re-ordered.  The rows are uploaded as they are and classified on the device.  A partition whose rows all fit
 * the mailbox vocabulary -
 *     at most one AppendResponse and one HeartbeatResponse per member slot (heads < JG_MAILBOX_NONE),
 *     at most one ClientRequest, and only if this node leads the partition,
 *     at most one Heartbeat and one AppendEntries, only if this node does NOT lead the partition (a leader
 *     answers them with a role change or not at all, leader.rs:200-208,263), from the same sender with the
 *     same term, the Heartbeat first (one answer word holds HeartbeatResponse, AppendResponse in that order),
 *     the AppendEntries' blocks a run (ids consecutive, each block's parent its predecessor, <= 254 blocks)
 * - is served in column form by the dense node tick (the HBM-bound kernels).  A column forgets the order of
 * the rows that filled it; what of that order can matter travels with the columns: which AppendResponses
 * arrived BEFORE the ClientRequest (they met the chain head before the append: chain.rs:197-202 holds them to
 * it, and what they committed precedes the Notify on fsm_tx), and for every partition the lag-space tick does
 * not serve (a HeartbeatResponse without the commit makes the leader replicate() on the progress as it is at
 * that moment, leader.rs:222-231; forged or far-behind heads) the arrival index of every row: those are
 * replayed one command at a time in arrival order.  Every row of any OTHER partition (votes, Timeout, Restart,
 * explicit Tick rows, duplicates, a ClientRequest at a non-leader, ...) is applied by the general state machine
 * exactly as jg_submit + jg_step would (stream order); that partition then takes part in the Tick like
 * everybody else.  `rows_general` counts those rows: a matter of cost, never of results.
 * A column handed out by jg_node_inbox_columns stands for that peer's answers AFTER the step's rows, per
 * partition HeartbeatResponse then AppendResponse, slots ascending.
 *
 * Outputs: rpc_tx - the Tick's Heartbeat / AppendEntries and the followers' answers as the mailbox
 * columns of jg_node_outbox (pinned host memory, one copy per column), everything else as rows through
 * jg_drain_messages.  Per partition the reference's emission order is: where an answer word is set - the
 * ClientRequest rows of the step (the queue the Heartbeat flushed, follower.rs:190-197), the word's
 * HeartbeatResponse, its AppendResponse, then the partition's other rows (what its Tick sent); elsewhere - the
 * partition's rows (a leader's extra AppendEntries of apply_heartbeat_response), then the Tick's words:
 * Heartbeat, AppendEntries by ascending slot.  fsm_tx - unlike the plain dense entry points, jg_step_node
 * QUEUES the FSM rows of its dense halves for jg_drain_applies, run-length encoded per partition and step, in
 * emission order: the Apply range (JG_FSM_APPLY_LEADER range(a..=b).skip(1)) committed by the
 * AppendResponses that arrived before the ClientRequest, the JG_FSM_NOTIFY {a = block id, b = the
 * ClientRequest's token}, the Apply range of the self-ack and the AppendResponses after it; a follower: one
 * JG_FSM_APPLY_FOLLOWER range(a..b) - consecutive ranges concatenate exactly (the progress heads only
 * grow).  Faults: jg_drain_faults. */
enum {
  JG_NODE_LEADER_HALF = 1u,   /* serve the partitions this node leads (jg_step_dense_leader)            */
  JG_NODE_FOLLOWER_HALF = 2u, /* serve the partitions it follows (jg_step_dense_follower)              */
  JG_NODE_TICK = 4u,          /* Command::Tick for every partition after its rows (server.rs:125)      */
  /* No synchronisation inside the call: it returns as soon as everything is ENQUEUED - uploads, classification, the
   * dense halves, the downloads of the outbox columns - and the engine's pinned input columns (jg_submit_reserve) are
   * a second set from then on, so that the next tick can be decoded while this one runs.  The dense halves of such a
   * step leave the partitions whose rows take the general path alone; the step is SETTLED by the next call that looks at
   * the engine (jg_node_outbox_view first of all; any step, drain, read or jg_sync): the row count has landed by then,
   * and if it is not zero those rows are applied and the halves come back for exactly those partitions - the same
   * results and the same record order as the synchronous step, one pass later.  Single-device engines or shards. */
  JG_NODE_ASYNC = 8u,
  /* The Tick's AppendEntries words as ONE word per partition where every addressee's word is the same (the steady state:
   * every follower acknowledged the same head): jg_node_outbox.aec[g] is that word - JG_NO_ACK: nothing for anybody -
   * or JG_AEC_INDIVIDUAL: the words are in jg_node_outbox.ae (valid for exactly those partitions: they differ by addressee,
   * or the partition's rows took the general path in this step), which is
   * NULL - and is not downloaded: 8 bytes per partition instead of 8 (R - 1) - when no partition of the step needs it
   * (jg_node_outbox_view fetches the rows when one does).  Single-device engines or shards.
   * WHICH partitions read JG_AEC_INDIVIDUAL is a matter of representation, not of results (the engine marks every
   * partition whose rows took the general path; the oracle library reports the common word wherever the R - 1 words
   * agree): compare aec / ae between backends only after expansion (BatchedRaft::expand_columns). */
  JG_NODE_COMMON_AE = 16u,
  /* A leader partition's fsm_tx rows of the step as one JG_FSM_LEADER_STEP row where the step appended a block and the
   * commit index is within 255 of it (24 bytes instead of 48 or 72; other partitions: the plain rows as before). */
  JG_NODE_FSM_FUSED = 32u,
  /* TWO STEPS IN FLIGHT (with JG_NODE_ASYNC; single-device engines or shards): the step keeps its outputs - the outbox
   * columns and every fsm_tx / rpc_tx row and fault it produced - in a set of its own until its outbox has been VIEWED,
   * and the next such step may be taken before that: while the device runs step t + 1 and its inputs travel up, step t's
   * outputs travel home (server.rs:103-165's channels are asynchronous: what a tick pushed on fsm_tx / rpc_tx is
   * consumed while the loop already takes the next messages).  With kept steps outstanding
   *   - jg_node_outbox_view serves the OLDEST one: it waits for that step's outputs only (never for the newer step),
   *     and makes exactly that step's rows - in the order a synchronous step would have queued them - what the next
   *     jg_drain_applies[_view] / jg_drain_messages[_view] / jg_drain_faults deliver; the pointers of the view stay
   *     valid until the second next kept step begins;
   *   - at most TWO may be outstanding: a third jg_step_node is JG_EINVAL, and so is any other stepping call
   *     (jg_step, a step without this flag, the dense entry points) - view the outboxes first;
   *   - reads (jg_read_state, jg_sync, counters) see the engine after the newest step.
   * The step's fsm rows come home behind its own kernels, the copy sized from the step before (no drain is issued by
   * the host, nothing waits for a count); rows of the general path, exceptional rows and faults - none in the steady
   * state - are collected when the outbox is viewed.  Same results, same rows, same order per step as without the flag. */
  JG_NODE_KEEP = 64u
};
#define JG_AEC_INDIVIDUAL 0xfffffffffffffffeull /* jg_node_outbox.aec: "see the rows of ae" (no JG_AE word: a range start stays below JG_MAILBOX_NONE) */
typedef struct jg_node_outbox { /* host pointers into the engine's pinned buffers; NULL: that half did not run */
  const jg_leader_beat* beat;  /* [G]    what a leader's followers read of its Tick (jg_leader_outbox.beat)   */
  const uint64_t* ae;          /* [R][G] JG_AE(from, n) per addressee slot, JG_NO_ACK: none (JG_NODE_COMMON_AE: see there) */
  const uint64_t* answer;      /* [G]    JG_ANSWER(AppendResponse.head, has_committed) to the partition's leader */
  const uint64_t* hb_commit;   /* [G]    HeartbeatResponse.commit (valid where the answer carries a response)  */
  uint64_t rows;               /* command rows the step took                                                   */
  uint64_t rows_general;       /* ... of which went through the general state machine                          */
  uint64_t bytes_h2d;          /* PCIe: uploaded for this step                                                 */
  uint64_t bytes_d2h;          /* PCIe: outbox columns downloaded for this step (drains not included)          */
  const uint64_t* aec;         /* [G]    JG_NODE_COMMON_AE: the partition's AppendEntries word for every addressee, else NULL */
} jg_node_outbox;
/* Column inbound for jg_step_node: a peer that is itself a batched engine ships its followers' answers as the
 * column it produced (jg_node_outbox.answer / .hb_commit) instead of two rows per partition.  The call hands out
 * where member slot `slot`'s column lives in the engine's pinned memory - *answer: [G] JG_ANSWER words (JG_NO_ACK:
 * nothing from that peer for the partition), *hb_commit: [G] HeartbeatResponse.commit (read only where the word
 * carries has_committed == 0); the caller fills them IN PLACE before the next jg_step_node, whose leader half applies
 * them for EVERY partition (after that step's general-path rows), exactly as if each word had been an
 * AppendResponse / HeartbeatResponse row of a column-form partition.  hb_commit == NULL: the commit is 0 for every
 * has_committed == 0 response of the column (not uploaded).  The hand-out covers ONE step.  `slot` must not be the
 * own slot of any partition (that word carries the append count), and a ROW of the same step that names the same
 * sender is an error (JG_EINVAL at the next synchronising call): a peer speaks columns or rows within one tick.
 * Single-device engines (or a shard's own handle). */
int jg_node_inbox_columns(jg_engine* e, uint32_t slot, uint64_t** answer, uint64_t** hb_commit);

/* Apply everything queued by jg_submit since the last step as described above (`flags`: JG_NODE_*; at
 * least one half).  Asynchronous like jg_step except for one synchronisation after the classification
 * (the number of general-path rows sizes that step's launch) - none at all with JG_NODE_ASYNC. */
int jg_step_node(jg_engine* e, uint64_t now_ms, uint32_t flags);
/* The mailbox columns the last jg_step_node produced: waits for them to land; the pointers stay valid
 * until the next jg_step_node.  On a multi-device engine the columns are the shards' concatenated.
 * (JG_NODE_KEEP: the OLDEST step whose outbox has not been viewed - see there.) */
int jg_node_outbox_view(jg_engine* e, jg_node_outbox* out);

/* ---- a closed loop of dense node ticks ------------------------------------------------------------
 * All R nodes of every partition in ONE process (one engine per node, e.g. the three brokers of
 * examples/multi-node in one runtime): the protocol round — leader half on nodes[lead], follower half
 * on every other node, each one's outbox columns being the others' inbox columns — driven from
 * inside the library, so that a round costs its launches and no per-call host overhead of the
 * caller's language.  The cluster owns the mailbox columns (in the memory of nodes[lead]'s device);
 * nothing synchronises with the host inside a round.  While the cluster exists, the nodes that share
 * nodes[lead]'s device issue ALL their work on its stream (the halves of a round are bandwidth-bound:
 * side by side they only slow each other down); nodes on other devices keep their streams and are
 * chained with events.
 * Equivalent, call for call, to jg_step_dense_leader on nodes[lead] followed by jg_step_dense_follower
 * (tick = 1) on the others (josefine::DenseCluster::round in josefine_amd/host/raft_handle.hpp).
 * Engines are borrowed: destroy the cluster BEFORE them (it gives them their streams back). */
typedef struct jg_dense_cluster jg_dense_cluster;
/* lead = JG_CLUSTER_ANY_LEADER: PER-PARTITION LEADERSHIP - every node leads the partitions it was elected for and
 * follows the others (what a josefine cluster looks like after its elections: candidate.rs:101-113 -> leader.rs:124-174).
 * The mailbox columns are then the CLUSTER's, indexed by group and slot: whoever leads group g reads row r of the
 * answers as slot r's answer and writes the beat and the AppendEntries words of g; every other node reads its
 * word, follows and answers into its own row.  A round is
 *   0. the claim: owner[g] = the lowest slot whose node is a healthy leader of g (JG_OWNER: nobody);
 *   1. every node's leader half over the groups it leads - the owner's exactly as jg_step_dense_leader (its own slot's
 *      word = the ClientRequests jg_dense_cluster_set_appends offers to whoever owns g), its Tick into the columns;
 *      a leader that is NOT the owner (two terms' leaders in one round) gets no inbox and no ClientRequest, and its
 *      Tick travels as rows (Heartbeat rows are delivered by the routed round's transport, AppendEntries rows stay
 *      queued for the host like every AppendEntries row);
 *   2. every node's follower half over the groups it does not lead, tick = 1, sender = the owner's NodeId - mail only
 *      where somebody else owns the group.
 * Equivalent, call for call, to that sequence of jg_step_dense_leader / jg_step_dense_follower calls with the inboxes
 * masked accordingly (tests/dense_node.py::AnyLeaderCluster states it in numpy).  Each node takes two steps per round.
 * The nodes must share a device; n_nodes <= 6; nodes[r] hosts replica slot r of every group. */
#define JG_CLUSTER_ANY_LEADER 0xFFFFFFFFu
int jg_dense_cluster_create(jg_engine* const* nodes, uint32_t n_nodes, uint32_t lead, jg_dense_cluster** out);
void jg_dense_cluster_destroy(jg_dense_cluster* c);
/* Per-cluster options (fixed before the cluster's first routed round; JG_EINVAL afterwards or for an unknown option).
 *   JG_CLUSTER_OPT_VOTE_WORDS (0 / 1): jg_dense_cluster_round_routed moves an ELECTION's traffic - a campaign's
 *     VoteRequest broadcasts (candidate.rs:24-45) and the VoteResponses they are answered with (follower.rs:219-246,
 *     candidate.rs:66-113) - as mailbox words per (partition, sender) read by a dense receiving half instead of as rows
 *     through the row transport, wherever EVERYTHING a node receives for a partition in a round is such traffic; what the
 *     nodes compute, emit and keep for the host is the row transport's, bit for bit (jg_route_stats.delivered counts rows
 *     only and is smaller).  Needs nodes that share the cluster's stream (one device) and R >= 2; otherwise ignored. */
enum { JG_CLUSTER_OPT_VOTE_WORDS = 1 };
int jg_dense_cluster_set_option(jg_dense_cluster* c, uint32_t option, uint64_t value);
/* ClientRequests every group appends per round (leader.rs:177-197): the same number for all groups,
 * or (per_group != NULL) one value per group from host memory.  (JG_CLUSTER_ANY_LEADER: offered to whoever owns the
 * group that round; a group nobody leads is offered nothing.) */
int jg_dense_cluster_set_appends(jg_dense_cluster* c, uint64_t uniform, const uint64_t* per_group);
/* No more ClientRequests for the n groups listed in DEVICE memory (ascending or not; duplicates allowed): their
 * offered count becomes 0 from the next round on - a client that has lost its partition's leader stops proposing
 * (what the reference makes of a re-elected leader is Q8: its first append panics, chain.rs:163).  Asynchronous on the
 * cluster's stream; the list must stay valid until then (jg_sync). */
int jg_dense_cluster_withdraw_appends(jg_dense_cluster* c, const uint32_t* groups_dev, uint32_t n);
/* ... and `per_round` ClientRequests per round for them from the next round on (0: withdraw) - the client proposes again to
 * partitions that were re-created (JG_CMD_RECREATE) and have a leader that can append.  Same conventions. */
int jg_dense_cluster_offer_appends(jg_dense_cluster* c, const uint32_t* groups_dev, uint32_t n, uint64_t per_round);
/* n_rounds protocol rounds at logical times now_ms, now_ms + dt_ms, ...; asynchronous (jg_sync the nodes).  Two or more
 * rounds are replayed as captured graphs (logical time and step numbers live in a device-resident clock that advances
 * itself), eight rounds to a graph where n_rounds allows; the results are those of n_rounds calls with one round each. */
int jg_dense_cluster_rounds(jg_dense_cluster* c, uint64_t now_ms, uint64_t dt_ms, uint32_t n_rounds);
/* The mailbox columns, for inspection: the leader's inbox / outbox as the structs above.  out->ae is a SNAPSHOT of the
 * last round's AppendEntries words, written out at this call (the cluster keeps ONE word per group where every follower's
 * is the same: JgLeaderNode::o_aec) and valid until the next round; with JG_CLUSTER_ANY_LEADER the rows of a group nobody
 * owns read JG_NO_ACK. */
int jg_dense_cluster_mailboxes(jg_dense_cluster* c, jg_leader_inbox* in, jg_leader_outbox* out);

/* One protocol round WITH a transport for everything outside the mailbox vocabulary — elections above
 * all: in the reference a VoteRequest leaves on rpc_tx, crosses tcp.rs and comes back out of the peer's
 * event loop as a Command (server.rs:127-137); here the rows the nodes queued during the round are
 * taken out of their undrained output ON THE DEVICE and become the addressees' first step of the next
 * routed round.  Per node and call:
 *   1. jg_step over the rows delivered by the previous call, then over inject[node] (a device batch as
 *      for jg_step_device_rows, or n == 0; NULL = nothing for anybody) — per group: the peers' rows in
 *      the order (phase of the round they were emitted in, emission index within the sender's step,
 *      sender slot), then the injected ones.  The phases are the steps the nodes take in lockstep: 1 = this
 *      step over the delivered rows, 2 = the injected rows, 3 = the leader half, 4 = the follower half of the
 *      dense round.  Every sender's stream arrives in its own order - all a network promises (one
 *      connection per peer pair, tcp.rs:87-170) - and the senders are interleaved: everybody's first row of a
 *      phase before anybody's second.  That interleaving is what lets an election of more than three nodes
 *      complete: a candidate broadcasts its VoteRequest once per peer (candidate.rs:30-37), a voter grants
 *      the first copy and refuses the rest, and a voter's later answer overwrites its earlier one
 *      (election.rs:33-35) - the quorum has to be seen among the first answers.  (Until ABI v8 the order was
 *      sender-major, each sender's rows back to back: five-node elections could not be won.);
 *   2. the dense round of jg_dense_cluster_rounds at now_ms, except that the ClientRequests of
 *      jg_dense_cluster_set_appends are offered only to groups nodes[lead] leads at that moment (a
 *      replica without a leader queues them, follower.rs:258-270: not a dense append; with per-partition
 *      leadership: to whoever owns the group);
 *   3. the transport: rows addressed to members (JG_TO_PEERS: all other members; JG_TO_PEER: to_id)
 *      are delivered, EXCEPT AppendEntries rows (the payload is the sender's block store) and
 *      ClientRequest rows (instructions to the host adapter about its request mirror): those, and
 *      everything addressed elsewhere, stay queued for jg_drain_messages, as do FSM rows.
 * A sender whose round left nothing for the host needs no drain (its output regions are recycled).
 * The nodes must share a device.  Synchronises with the host once per call (row counts).  Arguments are checked
 * before anything is launched; an error after the round has begun to consume the delivered rows (a HIP failure,
 * an internal inconsistency) leaves messages lost in flight: the cluster is marked failed and every later call
 * returns JG_EDEVICE - destroy it. */
typedef struct jg_route_stats {
  uint64_t delivered[JG_MAX_REPLICAS]; /* rows queued for node n's next routed round                   */
  uint64_t kept;                       /* message rows of this round left for jg_drain_messages        */
  uint64_t fsm_rows;                   /* FSM rows this round's steps queued (jg_drain_applies)        */
} jg_route_stats;
int jg_dense_cluster_round_routed(jg_dense_cluster* c, uint64_t now_ms, const jg_cmd_batch* inject /* [n_nodes] or NULL */,
                                  jg_route_stats* stats /* may be NULL */);

/* Batched Chain::compact (src/raft/chain.rs:239-253) as a pure function over
 * explicit (id,next) trees: tree t owns entries [off[t], off[t+1]); ids within a
 * tree need not be sorted.  removed[i] = 1 iff the walk removes entry i. */
int jg_chain_compact(jg_engine* e, size_t n_trees, const uint64_t* off /*[n_trees+1]*/,
                     const uint64_t* ids, const uint64_t* nexts, const uint64_t* commits /*[n_trees]*/,
                     uint8_t* removed /* [off[n_trees]] out */);

/* Chain::compact (src/raft/chain.rs:239-253, incl. its quirk Q7: the parent pointer of a REMOVED
 * block is followed too) on the resident chain of every healthy group: the blocks below the
 * commit index that the walk removes leave the engine's id set, and are queued as rows for the
 * host to delete from its block store (jg_drain_compacted: group ascending, ids descending = the
 * order of the walk).  *n_removed (optional) = rows this call queued.  The reference never calls
 * compact() outside its test; a host that wants its dead branches gone calls this between steps. */
typedef struct jg_compact_row {
  uint32_t group;
  uint32_t pad;
  uint64_t id; /* BlockId removed from the chain of `group` */
} jg_compact_row;
int jg_chain_compact_resident(jg_engine* e, size_t* n_removed);
int jg_drain_compacted(jg_engine* e, jg_compact_row* out, size_t cap, size_t* n);

int jg_sync(jg_engine* e);

/* Device-side ordering between two engines of one process: everything `waiter` enqueues after
 * this call runs after everything `signal` has enqueued so far (hipEventRecord on signal's
 * stream + hipStreamWaitEvent on waiter's; no host synchronisation).  What chains the dense
 * mailboxes of several engines — one engine's outbox columns are another's inbox. */
int jg_stream_wait(jg_engine* waiter, jg_engine* signal);

/* Drains.  Each returns the queued rows (per-group emission order identical to the
 * reference; groups in ascending order within one step) and clears the queue.
 * With out == NULL only *n is set. */
int jg_drain_messages(jg_engine* e, jg_msg_row* out, size_t cap, size_t* n);
int jg_drain_applies(jg_engine* e, jg_fsm_row* out, size_t cap, size_t* n);
int jg_drain_faults(jg_engine* e, jg_fault_row* out, size_t cap, size_t* n);
/* Zero-copy drains: *rows points at the engine's pinned host queue (filled by one
 * asynchronous device-to-host copy), valid until the next drain call on the SAME queue
 * (drains of the other queues, faults, steps and reads in between leave them alone; step
 * results are not affected: the rows count as drained).  What a Rust adapter iterates to
 * re-emit on rpc_tx / fsm_tx without an intermediate Vec. */
int jg_drain_messages_view(jg_engine* e, const jg_msg_row** rows, size_t* n);
int jg_drain_applies_view(jg_engine* e, const jg_fsm_row** rows, size_t* n);

/* Pipelined drains.  jg_drain_prefetch marks a point in the step stream and starts, without
 * blocking, the compaction and the transfer to the host queues of every row that steps before
 * the point have produced — on a second HIP stream behind an event, driven by the engine's own
 * drain thread, while the caller keeps stepping.  From the first call on the engine is PIPELINED:
 *   - jg_drain_prefetch never blocks: while a batch is still in transfer it starts nothing (the
 *     next call covers more steps);
 *   - the jg_drain_* calls never block and never synchronise with the step stream: they deliver the
 *     rows of the batches that have landed (same order as ever: steps in order, groups ascending
 *     within a step), possibly none;
 *   - jg_drain_wait blocks until the batch in transfer (if any) has landed - the back-pressure of a
 *     caller that does not want to run more than one batch ahead of its own output;
 *   - jg_drain_flush blocks until everything stepped so far has landed (then drain).
 * The pattern for a caller that forwards messages every N ticks (what server::event_loop does with
 * rpc_tx for one group, src/raft/server.rs:125-159):
 *     every N ticks: jg_drain_*_view (whatever has landed) ... jg_drain_prefetch (next batch)
 * so that PCIe time and host time of one batch overlap the device time of the following ticks. */
int jg_drain_prefetch(jg_engine* e);
int jg_drain_wait(jg_engine* e);
int jg_drain_flush(jg_engine* e);

/* Copy one state column for groups [g0, g0+n) to host memory (element type per
 * JG_FIELD_* above).  `replica` selects the slot for JG_FIELD_MATCH. */
int jg_read_state(jg_engine* e, int field, uint32_t replica, void* out, uint32_t g0, uint32_t n);

/* Counters since creation: [0] commands applied, [1] quorum decisions
 * (Leader::commit evaluations + election_status evaluations),
 * [2] group-steps of the dense path, [3] kernel launches. */
int jg_get_counters(jg_engine* e, uint64_t out[4]);

/* ---- device-resident helpers for benchmarks ---------------------------------
 * The engine owns its stream; these let a caller keep inputs in HBM and time
 * the stream with HIP events without linking the HIP runtime itself. */
int jg_device_alloc(jg_engine* e, size_t bytes, void** dev_ptr);
int jg_device_free(jg_engine* e, void* dev_ptr);
int jg_device_upload(jg_engine* e, void* dev_dst, const void* host_src, size_t bytes);
int jg_device_download(jg_engine* e, void* host_dst, const void* dev_src, size_t bytes);
int jg_timer_start(jg_engine* e);            /* hipEventRecord on the engine stream */
int jg_timer_stop(jg_engine* e, float* ms);  /* record + synchronize + elapsed      */

/* Synthetic AppendEntries-ack stream generator,
 * counter-based so any shard regenerates its slice: fills one dense [R][G] tick
 * on the device from (seed, tick, global group, replica) and the generator's own
 * follower model kept in `sim` ([R][G] u64, zero-initialised by the caller).
 * mode 0: steady state (#3/#4) — 1 append per tick, every follower acks the
 *         leader head of the previous tick;
 * mode 1: ragged (#2) — a in {0,1,2} appends, follower ack = min(leader_head,
 *         prev_ack + U{0..MAX_INFLIGHT}), 5 % dropped, 5 % stale duplicates. */
int jg_synth_fill_acks_device(jg_engine* e, uint32_t mode, uint64_t tick, uint64_t* sim_dev,
                              uint64_t* acks_dev);

/* Measurement aid: HIP event pairs around the dense tick kernel itself (k_leader_tick_dense /
 * _n / k_leader_node_tick — not the k_dense_slow launch that may follow it), recorded on the
 * engine's stream for every dense step while enabled; jg_kernel_timing_read synchronises and
 * returns the average over the most recent launches (a ring of 256).  What bench.py prices the
 * roofline with when a step is more than one kernel (configs[4], the closed loop).
 * enable: 0 off, 1 every dense step, N > 1 every N-th (a pair of event records costs the stream a few
 * microseconds: a tick of three small kernels is measurably longer with every launch timed). */
int jg_kernel_timing(jg_engine* e, int enable);
int jg_kernel_timing_read(jg_engine* e, float* avg_us, uint32_t* n_launches);

/* Measurement aid (bench.py "roofline.stream_ceiling"): time a plain streaming kernel with the
 * byte profile of the dense leader tick for this engine's (G, R) — per group R 8-byte reads from
 * a rotating ack-sized buffer set larger than the Infinity Cache (non-temporal, like the ack
 * stream), two resident 8-byte columns and one 4-byte column read, one 8-byte column written —
 * and no Raft logic at all.  Same grid, workgroup size and stream as jg_step_dense_acks_device.
 * *avg_us = average launch duration over `iters` launches (HIP events).  What a launch of this
 * shape costs on the machine at hand: the practical ceiling next to the 8 TB/s spec peak. */
int jg_calibrate_stream(jg_engine* e, uint32_t iters, float* avg_us);

const char* jg_last_error(void);
uint32_t jg_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JOSEFINE_GPU_H */
