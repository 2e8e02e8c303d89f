"""ctypes view of include/josefine_gpu.h.

The same table binds any shared library that exports the ABI under a symbol
prefix: the shipped HIP engine uses ``jg_``; the test-suite binds its CPU
checker (prefix ``jo_``) through the very same table so both are driven by
identical code.  Nothing in this package loads that checker.
"""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 9
CLUSTER_ANY_LEADER = 0xFFFFFFFF
CLUSTER_OPT_VOTE_WORDS = 1
MAX_REPLICAS = 8
CHAIN_WINDOW = 8
MAX_INFLIGHT = 5
MAX_DEVICES = 16
FOREIGN_VOTERS = 8
MAX_DENSE_APPENDS = 1 << 20
NO_ACK = 0xFFFFFFFFFFFFFFFF
AEC_INDIVIDUAL = 0xFFFFFFFFFFFFFFFE  # jg_node_outbox.aec: the partition's AppendEntries words differ by addressee (see .ae)

OK, EINVAL, ENOMEM, EDEVICE, ECAPACITY = 0, -1, -2, -3, -4

ROLE_FOLLOWER, ROLE_CANDIDATE, ROLE_LEADER = 0, 1, 2

(CMD_TICK, CMD_PROPOSE, CMD_VOTE_REQUEST, CMD_VOTE_RESPONSE, CMD_APPEND_ENTRIES,
 CMD_APPEND_RESPONSE, CMD_HEARTBEAT, CMD_HEARTBEAT_RESPONSE, CMD_TIMEOUT, CMD_NOOP,
 CMD_CLIENT_REQUEST, CMD_CLIENT_RESPONSE, CMD_RESTART, CMD_RECREATE) = range(14)

TO_PEERS, TO_PEER, TO_LOCAL, TO_CLIENT, TO_QUEUE = range(5)
QUEUE_FLUSH, QUEUE_DROP = 1, 2

FAULT_NONE = 0
FAULT_LEADER_TERM_UNIMPLEMENTED = 1
FAULT_APPEND_ID_NOT_ABOVE_HEAD = 2
FAULT_PROGRESS_UNKNOWN_NODE = 3
FAULT_COMMIT_MISSING_BLOCK = 4
FAULT_EXTEND_MISSING_PARENT = 5
FAULT_FOLLOWER_STALE_LEADER = 6
FAULT_CANDIDATE_TICK_ELECTED = 7
FAULT_RANGE_HIT_COMMIT_KEY = 8
FAULT_ENGINE_WINDOW_OVERFLOW = 128
FAULT_ENGINE_FOREIGN_VOTER = 129
FAULT_ENGINE_DENSE_NONLEADER = 131
FAULT_ENGINE_DENSE_APPENDS = 132
FAULT_ENGINE_MAILBOX_RANGE = 133

CFG_SEPARATE_COMMIT_KEY = 1
CFG_FLAT_ROW_PASSES = 2  # jg_step_node's row passes untiled (the tiled ones' statement; an A/B)
NODE_LEADER_HALF, NODE_FOLLOWER_HALF, NODE_TICK, NODE_ASYNC, NODE_COMMON_AE, NODE_FSM_FUSED, NODE_KEEP = 1, 2, 4, 8, 16, 32, 64

FSM_APPLY_LEADER, FSM_APPLY_FOLLOWER, FSM_NOTIFY, FSM_LEADER_STEP = 0, 1, 2, 3

(FIELD_TERM, FIELD_VOTED_FOR, FIELD_HAS_VOTED, FIELD_ROLE, FIELD_COMMIT, FIELD_HEAD,
 FIELD_ID_GEN, FIELD_MATCH, FIELD_REPL_STATE, FIELD_VOTE_SEEN, FIELD_VOTE_GRANTED,
 FIELD_FAULT, FIELD_LEADER_ID, FIELD_HAS_LEADER, FIELD_ELECTION_TIME,
 FIELD_ELECTION_TIMEOUT, FIELD_HEARTBEAT_TIME, FIELD_QUEUED_REQS, FIELD_SELF_SLOT) = range(19)

# numpy dtype per readable column
FIELD_DTYPES = {
    FIELD_TERM: "u8", FIELD_VOTED_FOR: "u4", FIELD_HAS_VOTED: "u1", FIELD_ROLE: "u1",
    FIELD_COMMIT: "u8", FIELD_HEAD: "u8", FIELD_ID_GEN: "u8", FIELD_MATCH: "u8",
    FIELD_REPL_STATE: "u1", FIELD_VOTE_SEEN: "u1", FIELD_VOTE_GRANTED: "u1", FIELD_FAULT: "u1",
    FIELD_LEADER_ID: "u4", FIELD_HAS_LEADER: "u1", FIELD_ELECTION_TIME: "u8",
    FIELD_ELECTION_TIMEOUT: "u4", FIELD_HEARTBEAT_TIME: "u8", FIELD_QUEUED_REQS: "u4",
    FIELD_SELF_SLOT: "u1",
}
FIELD_NAMES = {
    "term": FIELD_TERM, "voted_for": FIELD_VOTED_FOR, "has_voted": FIELD_HAS_VOTED,
    "role": FIELD_ROLE, "commit": FIELD_COMMIT, "head": FIELD_HEAD, "id_gen": FIELD_ID_GEN,
    "match": FIELD_MATCH, "repl_state": FIELD_REPL_STATE, "vote_seen": FIELD_VOTE_SEEN,
    "vote_granted": FIELD_VOTE_GRANTED, "fault": FIELD_FAULT, "leader_id": FIELD_LEADER_ID,
    "has_leader": FIELD_HAS_LEADER, "election_time": FIELD_ELECTION_TIME,
    "election_timeout": FIELD_ELECTION_TIMEOUT, "heartbeat_time": FIELD_HEARTBEAT_TIME,
    "queued_reqs": FIELD_QUEUED_REQS, "self_slot": FIELD_SELF_SLOT,
}


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_groups", C.c_uint32),
        ("n_replicas", C.c_uint32),
        ("node_ids", C.c_uint32 * MAX_REPLICAS),
        ("device_id", C.c_int32),
        ("heartbeat_timeout_ms", C.c_uint32),
        ("election_timeout_min_ms", C.c_uint32),
        ("election_timeout_max_ms", C.c_uint32),
        ("seed", C.c_uint64),
        ("group_base", C.c_uint64),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
        ("n_devices", C.c_uint32),
        ("device_ids", C.c_int32 * MAX_DEVICES),
    ]


class ShardInfo(C.Structure):
    _fields_ = [("engine", C.c_void_p), ("device_id", C.c_int32), ("group_lo", C.c_uint32),
                ("n_groups", C.c_uint32), ("reserved", C.c_uint32)]


class RouteStats(C.Structure):
    """jg_route_stats."""
    _fields_ = [("delivered", C.c_uint64 * 8), ("kept", C.c_uint64), ("fsm_rows", C.c_uint64)]


class NodeOutbox(C.Structure):
    """jg_node_outbox."""
    _fields_ = [("beat", C.c_void_p), ("ae", C.c_void_p), ("answer", C.c_void_p), ("hb_commit", C.c_void_p),
                ("rows", C.c_uint64), ("rows_general", C.c_uint64), ("bytes_h2d", C.c_uint64), ("bytes_d2h", C.c_uint64),
                ("aec", C.c_void_p)]


class CmdCols(C.Structure):
    """jg_cmd_cols."""
    _fields_ = [(k, C.c_void_p) for k in ("kind", "group", "from_", "term", "id", "aux", "flag", "blk_id", "blk_next")]


COL_FROM, COL_TERM, COL_AUX, COL_FLAG, COL_UNCHECKED, COL_UPLOAD_NOW, COL_PACKED_KIND, COL_ID32 = 1, 2, 4, 8, 16, 32, 64, 128


class CmdBatch(C.Structure):
    _fields_ = [
        ("n", C.c_size_t),
        ("kind", C.c_void_p),
        ("group", C.c_void_p),
        ("from_", C.c_void_p),
        ("term", C.c_void_p),
        ("id", C.c_void_p),
        ("aux", C.c_void_p),
        ("flag", C.c_void_p),
        ("n_blocks", C.c_size_t),
        ("blk_id", C.c_void_p),
        ("blk_next", C.c_void_p),
    ]


AE_NONE = 0xFF
HB_NONE = 0xFF


class LeaderInbox(C.Structure):
    _fields_ = [("answers", C.c_void_p), ("hbr_commit", C.c_void_p)]


class LeaderOutbox(C.Structure):
    _fields_ = [("beat", C.c_void_p), ("ae", C.c_void_p)]


class FollowerInbox(C.Structure):
    _fields_ = [("leader", C.c_void_p), ("leader_id", C.c_uint32), ("reserved", C.c_uint32),
                ("beat", C.c_void_p), ("ae", C.c_void_p)]


class FollowerOutbox(C.Structure):
    _fields_ = [("answer", C.c_void_p), ("hb_commit", C.c_void_p)]


# mailbox words (josefine_gpu.h: JG_ANSWER / JG_AE)
MAILBOX_NONE = (1 << 56) - 1


def pack_answers(ack_head, hb_has):
    """JG_ANSWER: AppendResponse.head (NO_ACK: none) and the HeartbeatResponse code per entry."""
    import numpy as np
    a = np.asarray(ack_head, dtype=np.uint64)
    field = np.where(a == np.uint64(NO_ACK), np.uint64(MAILBOX_NONE), a)
    # (a head that does not fit 56 bits cannot be put on the wire: the engines fault where they produce one)
    return (field << np.uint64(8)) | np.asarray(hb_has, dtype=np.uint64)


def unpack_answers(words):
    import numpy as np
    w = np.asarray(words, dtype=np.uint64)
    field = w >> np.uint64(8)
    return np.where(field == np.uint64(MAILBOX_NONE), np.uint64(NO_ACK), field), (w & np.uint64(0xFF)).astype(np.uint8)


def pack_ae(ae_from, ae_n):
    import numpy as np
    n = np.asarray(ae_n, dtype=np.uint64)
    return np.where(n == np.uint64(AE_NONE), np.uint64(NO_ACK), (np.asarray(ae_from, dtype=np.uint64) << np.uint64(8)) | n)


def unpack_ae(words):
    import numpy as np
    w = np.asarray(words, dtype=np.uint64)
    n = (w & np.uint64(0xFF)).astype(np.uint8)
    return np.where(n == AE_NONE, np.uint64(0), w >> np.uint64(8)), n


# numpy structured dtypes matching jg_msg_row / jg_fsm_row / jg_fault_row
MSG_DTYPE = [("group", "<u4"), ("kind", "u1"), ("to_kind", "u1"), ("flag", "u1"), ("pad", "u1"),
             ("to_id", "<u4"), ("from", "<u4"), ("term", "<u8"), ("id", "<u8"), ("aux", "<u8")]
FSM_DTYPE = [("group", "<u4"), ("kind", "u1"), ("pad", "u1", (3,)), ("a", "<u8"), ("b", "<u8")]
FAULT_DTYPE = [("group", "<u4"), ("code", "<u4")]
COMPACT_DTYPE = [("group", "<u4"), ("pad", "<u4"), ("id", "<u8")]

_P = C.c_void_p


class Api:
    """Function table for one loaded library."""

    # name -> (restype, argtypes); names without prefix
    _PROTOS = {
        "engine_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
        "engine_destroy": (None, [_P]),
        "set_self_slots": (C.c_int, [_P, _P]),
        "submit": (C.c_int, [_P, C.POINTER(CmdBatch)]),
        "step": (C.c_int, [_P, C.c_uint64]),
        "step_dense_acks": (C.c_int, [_P, _P]),
        "step_dense_leader": (C.c_int, [_P, C.c_uint64, C.POINTER(LeaderInbox), C.POINTER(LeaderOutbox)]),
        "step_dense_follower": (C.c_int, [_P, C.c_uint64, C.POINTER(FollowerInbox), C.POINTER(FollowerOutbox), C.c_int]),
        "chain_compact": (C.c_int, [_P, C.c_size_t, _P, _P, _P, _P, _P]),
        "chain_compact_resident": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
        "drain_compacted": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
        "drain_messages": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
        "drain_applies": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
        "drain_faults": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
        "read_state": (C.c_int, [_P, C.c_int, C.c_uint32, _P, C.c_uint32, C.c_uint32]),
        "get_counters": (C.c_int, [_P, C.POINTER(C.c_uint64 * 4)]),
        "last_error": (C.c_char_p, []),
        "abi_version": (C.c_uint32, []),
        "step_node": (C.c_int, [_P, C.c_uint64, C.c_uint32]),
        "node_outbox_view": (C.c_int, [_P, C.POINTER(NodeOutbox)]),
        "node_inbox_columns": (C.c_int, [_P, C.c_uint32, C.POINTER(_P), C.POINTER(_P)]),
    }
    # only the device engine has these
    _DEVICE_PROTOS = {
        "shard_count": (C.c_uint32, [_P]),
        "get_shard": (C.c_int, [_P, C.c_uint32, C.POINTER(ShardInfo)]),
        "step_dense_acks_shards": (C.c_int, [_P, C.POINTER(_P), C.c_uint32]),
        "step_device_rows": (C.c_int, [_P, C.POINTER(CmdBatch), C.c_uint64]),
        "step_dense_acks_device": (C.c_int, [_P, _P]),
        "step_dense_acks_device_n": (C.c_int, [_P, _P, C.c_uint32]),
        "sync": (C.c_int, [_P]),
        "stream_wait": (C.c_int, [_P, _P]),
        "drain_prefetch": (C.c_int, [_P]),
        "drain_flush": (C.c_int, [_P]),
        "drain_wait": (C.c_int, [_P]),
        "drain_messages_view": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
        "drain_applies_view": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
        "device_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
        "device_free": (C.c_int, [_P, _P]),
        "device_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
        "device_download": (C.c_int, [_P, _P, _P, C.c_size_t]),
        "timer_start": (C.c_int, [_P]),
        "timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
        "synth_fill_acks_device": (C.c_int, [_P, C.c_uint32, C.c_uint64, _P, _P]),
        "calibrate_stream": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_float)]),
        "submit_reserve": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.POINTER(CmdCols)]),
        "submit_commit": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_uint32]),
        "dense_cluster_create": (C.c_int, [C.POINTER(_P), C.c_uint32, C.c_uint32, C.POINTER(_P)]),
        "dense_cluster_destroy": (None, [_P]),
        "dense_cluster_set_option": (C.c_int, [_P, C.c_uint32, C.c_uint64]),
        "dense_cluster_set_appends": (C.c_int, [_P, C.c_uint64, _P]),
        "dense_cluster_withdraw_appends": (C.c_int, [_P, _P, C.c_uint32]),
        "dense_cluster_offer_appends": (C.c_int, [_P, _P, C.c_uint32, C.c_uint64]),
        "dense_cluster_rounds": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32]),
        "dense_cluster_mailboxes": (C.c_int, [_P, C.POINTER(LeaderInbox), C.POINTER(LeaderOutbox)]),
        "dense_cluster_round_routed": (C.c_int, [_P, C.c_uint64, C.POINTER(CmdBatch), C.POINTER(RouteStats)]),
        "kernel_timing": (C.c_int, [_P, C.c_int]),
        "kernel_timing_read": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    }
    # only the oracle has these
    _ORACLE_PROTOS = {
        "set_threads": (C.c_int, [_P, C.c_uint]),
        "synth_fill_acks": (C.c_int, [_P, C.c_uint32, C.c_uint64, _P, _P]),
    }

    def __init__(self, path: str, prefix: str):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL if False else C.RTLD_LOCAL)
        for table, required in ((self._PROTOS, True), (self._DEVICE_PROTOS, False), (self._ORACLE_PROTOS, False)):
            for name, (res, args) in table.items():
                sym = prefix + name
                try:
                    fn = getattr(self.lib, sym)
                except AttributeError:
                    if required:
                        raise ImportError(f"{path} does not export {sym}")
                    continue
                fn.restype = res
                fn.argtypes = args
                setattr(self, name, fn)

    def exported(self, name: str) -> bool:
        return hasattr(self.lib, self.prefix + name)

    def error(self) -> str:
        msg = self.last_error()
        return msg.decode() if msg else ""


# Every symbol include/josefine_gpu.h declares (checked by the CPU test-suite).
HEADER_SYMBOLS = [
    "jg_engine_create", "jg_engine_destroy", "jg_shard_count", "jg_get_shard", "jg_step_dense_acks_shards", "jg_set_self_slots", "jg_submit", "jg_step", "jg_step_device_rows",
    "jg_step_dense_acks", "jg_step_dense_acks_device", "jg_step_dense_acks_device_n",
    "jg_step_dense_leader", "jg_step_dense_follower", "jg_chain_compact", "jg_chain_compact_resident", "jg_drain_compacted", "jg_sync", "jg_stream_wait",
    "jg_drain_messages", "jg_drain_applies", "jg_drain_faults", "jg_drain_messages_view", "jg_drain_applies_view", "jg_drain_prefetch", "jg_drain_flush", "jg_drain_wait", "jg_read_state", "jg_get_counters",
    "jg_device_alloc", "jg_device_free", "jg_device_upload", "jg_device_download",
    "jg_timer_start", "jg_timer_stop", "jg_synth_fill_acks_device", "jg_calibrate_stream", "jg_dense_cluster_create", "jg_dense_cluster_destroy", "jg_dense_cluster_set_option", "jg_dense_cluster_set_appends", "jg_dense_cluster_withdraw_appends", "jg_dense_cluster_offer_appends", "jg_dense_cluster_rounds", "jg_dense_cluster_mailboxes", "jg_dense_cluster_round_routed", "jg_kernel_timing", "jg_kernel_timing_read", "jg_last_error", "jg_abi_version",
    "jg_step_node", "jg_node_outbox_view", "jg_submit_reserve", "jg_submit_commit", "jg_node_inbox_columns",
]
