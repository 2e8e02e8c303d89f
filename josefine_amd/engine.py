"""Host-side mirror of josefine's Raft driver surface over the C ABI.

The reference drives one Raft group through

    trait Apply { fn apply(self, cmd: Command) -> Result<RaftHandle>; }
                                        (src/raft/mod.rs:483-489)

with `Command` (src/raft/mod.rs:160-227) as the input vocabulary and two
output channels, `rpc_tx` / `fsm_tx` (src/raft/mod.rs:337-340).  `BatchedRaft`
is the same surface for N groups at once: `Command` keeps the reference's
variant and field names, `RaftHandle` keeps `apply` / `is_leader` / … , and the
channels become `drain_messages()` / `drain_applies()`.

There is no CPU implementation behind this class: it binds
`josefine_amd/csrc/libjosefine_gpu.so` (HIP, gfx950) and raises if that
library is missing or no device is usable.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _capi as capi

_LIB_PATH = os.environ.get("JOSEFINE_GPU_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                               "libjosefine_gpu.so")
_device_api: Optional[capi.Api] = None


class EngineError(RuntimeError):
    """A C-ABI call returned a negative status (anyhow::Error in the reference)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"josefine engine error {status}: {message}")
        self.status = status


def device_api() -> capi.Api:
    """Load the HIP engine library; never falls back to anything else."""
    global _device_api
    if _device_api is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        api = capi.Api(_LIB_PATH, "jg_")
        # (tests/host_device.py compiles the engine for the host against an emulated runtime and marks that build: it is
        # test infrastructure, loaded by the tests' child processes only - never a way to run the product without a GPU)
        if hasattr(api.lib, "jg_emulated_device") and os.environ.get("JG_EMULATED_DEVICE") != "1":
            raise ImportError(f"{_LIB_PATH} is the tests' emulated-device build, not the gfx950 engine. There is no CPU fallback.")
        _device_api = api
    return _device_api


@dataclass
class Command:
    """enum Command (src/raft/mod.rs:160-227), flattened to the batch columns."""

    kind: int
    from_: int = 0
    term: int = 0
    id: int = 0
    aux: int = 0
    flag: int = 0
    blocks: List[Tuple[int, int]] = field(default_factory=list)  # Vec<Block> as (id, next)

    # constructors named after the reference variants and their fields
    @staticmethod
    def Tick() -> "Command":
        return Command(capi.CMD_TICK)

    @staticmethod
    def Propose() -> "Command":
        return Command(capi.CMD_PROPOSE)

    @staticmethod
    def VoteRequest(term: int, candidate_id: int, last_term: int, head: int) -> "Command":
        return Command(capi.CMD_VOTE_REQUEST, from_=candidate_id, term=term, id=head, aux=last_term)

    @staticmethod
    def VoteResponse(term: int, from_: int, granted: bool) -> "Command":
        return Command(capi.CMD_VOTE_RESPONSE, from_=from_, term=term, flag=int(granted))

    @staticmethod
    def AppendEntries(term: int, leader_id: int, blocks: Sequence[Tuple[int, int]]) -> "Command":
        return Command(capi.CMD_APPEND_ENTRIES, from_=leader_id, term=term, blocks=list(blocks))

    @staticmethod
    def AppendResponse(node_id: int, term: int, head: int, success: bool = True) -> "Command":
        return Command(capi.CMD_APPEND_RESPONSE, from_=node_id, term=term, id=head, flag=int(success))

    @staticmethod
    def Heartbeat(term: int, commit: int, leader_id: int) -> "Command":
        return Command(capi.CMD_HEARTBEAT, from_=leader_id, term=term, id=commit)

    @staticmethod
    def HeartbeatResponse(commit: int, has_committed: bool) -> "Command":
        return Command(capi.CMD_HEARTBEAT_RESPONSE, id=commit, flag=int(has_committed))

    @staticmethod
    def Timeout() -> "Command":
        return Command(capi.CMD_TIMEOUT)

    @staticmethod
    def Noop() -> "Command":
        return Command(capi.CMD_NOOP)

    @staticmethod
    def ClientRequest(id: int = 0) -> "Command":
        return Command(capi.CMD_CLIENT_REQUEST, id=id)

    @staticmethod
    def ClientResponse(id: int = 0) -> "Command":
        return Command(capi.CMD_CLIENT_RESPONSE, id=id)

    @staticmethod
    def Restart() -> "Command":
        """Engine op: process restart (Raft::new + Chain::new on the persisted tree)."""
        return Command(capi.CMD_RESTART)

    @staticmethod
    def Recreate() -> "Command":
        """Engine op: the replica of a partition that was re-created (Raft::new + Chain::new on an EMPTY directory)."""
        return Command(capi.CMD_RECREATE)


def expand_fsm_rows(rows: np.ndarray) -> np.ndarray:
    """FSM rows (capi.FSM_DTYPE) with every FSM_LEADER_STEP row (jg_step_node with JG_NODE_FSM_FUSED) replaced by the rows
    it stands for, in order: Apply {c0, c1} if c1 != c0, Notify {a, b}, Apply {c1, c2} if c2 != c1 (c_k = a - pad[k])."""
    rows = np.asarray(rows)
    fused = rows["kind"] == capi.FSM_LEADER_STEP
    if not fused.any():
        return rows
    a = rows["a"].astype(np.uint64)
    c = [a - rows["pad"][:, k].astype(np.uint64) for k in range(3)]
    pre, post = fused & (c[1] != c[0]), fused & (c[2] != c[1])
    count = np.where(fused, 1 + pre.astype(np.int64) + post.astype(np.int64), 1)
    start = np.concatenate([[0], np.cumsum(count)[:-1]])
    out = np.zeros(int(count.sum()), dtype=rows.dtype)
    out[start[~fused]] = rows[~fused]
    i = np.nonzero(fused)[0]
    at = start[i]
    p = pre[i]
    out["group"][at[p]], out["kind"][at[p]], out["a"][at[p]], out["b"][at[p]] = rows["group"][i[p]], capi.FSM_APPLY_LEADER, c[0][i[p]], c[1][i[p]]
    at = at + p
    out["group"][at], out["kind"][at], out["a"][at], out["b"][at] = rows["group"][i], capi.FSM_NOTIFY, a[i], rows["b"][i]
    at = at + 1
    q = post[i]
    out["group"][at[q]], out["kind"][at[q]], out["a"][at[q]], out["b"][at[q]] = rows["group"][i[q]], capi.FSM_APPLY_LEADER, c[1][i[q]], c[2][i[q]]
    return out


class BatchedRaft:
    """N independent Raft node instances behind one engine handle."""

    def __init__(self, n_groups: int, n_replicas: int = 1, node_ids: Optional[Sequence[int]] = None,
                 self_slots: Optional[Sequence[int]] = None, seed: int = 0, device_id: int = 0,
                 group_base: int = 0, flags: int = 0, heartbeat_timeout_ms: int = 100,
                 election_timeout_ms: Tuple[int, int] = (500, 1000), api: Optional[capi.Api] = None,
                 device_ids: Optional[Sequence[int]] = None):
        """`device_ids` (HIP engine only): shard the groups over these devices behind this one
        handle (contiguous ownership; a device may be listed several times) — jg_config.n_devices."""
        self.api = api if api is not None else device_api()
        self.G, self.R = int(n_groups), int(n_replicas)
        if node_ids is None:
            node_ids = list(range(1, self.R + 1))  # examples/multi-node/node-*.toml: ids 1..R
        cfg = capi.Config()
        cfg.abi_version = capi.ABI_VERSION
        cfg.n_groups = self.G
        cfg.n_replicas = self.R
        for r, nid in enumerate(list(node_ids)[:capi.MAX_REPLICAS]):
            cfg.node_ids[r] = int(nid)
        cfg.device_id = device_id
        if device_ids is not None:
            assert 1 <= len(device_ids) <= capi.MAX_DEVICES
            cfg.n_devices = len(device_ids)
            for d, dev in enumerate(device_ids):
                cfg.device_ids[d] = int(dev)
        cfg.heartbeat_timeout_ms = heartbeat_timeout_ms
        cfg.election_timeout_min_ms, cfg.election_timeout_max_ms = election_timeout_ms
        cfg.seed = seed
        cfg.group_base = group_base
        cfg.flags = flags
        self.node_ids = [int(x) for x in node_ids]
        self.cfg = cfg
        h = C.c_void_p()
        self._check(self.api.engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._pending: List[Tuple[int, Command]] = []
        if self_slots is not None:
            s = np.ascontiguousarray(self_slots, dtype=np.uint8)
            assert s.shape == (self.G,)
            self._check(self.api.set_self_slots(self._h, s.ctypes.data))

    # -- lifecycle -----------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self.api.engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status: int) -> None:
        if status != capi.OK:
            raise EngineError(status, self.api.error())

    # -- shards of a multi-device engine -------------------------------------
    @property
    def n_shards(self) -> int:
        return int(self.api.shard_count(self._h)) if hasattr(self.api, "shard_count") else 1

    def shard(self, d: int) -> "Shard":
        """Shard d's own single-device engine (jg_get_shard): what the device-pointer entry
        points are called on."""
        info = capi.ShardInfo()
        self._check(self.api.get_shard(self._h, d, C.byref(info)))
        return Shard(self, info)

    def step_dense_acks_shards(self, ptrs: Sequence[int], n_ticks: int = 1) -> None:
        """jg_step_dense_acks_shards: ptrs[d] = shard d's device-resident [n_ticks][R][G_d] block."""
        arr = (C.c_void_p * len(ptrs))(*[C.c_void_p(int(p)) for p in ptrs])
        self._check(self.api.step_dense_acks_shards(self._h, arr, int(n_ticks)))

    # -- input ---------------------------------------------------------------
    def submit(self, group: int, cmd: Command) -> None:
        """Queue `cmd` for `group`; applied at the next step() in submit order."""
        self._pending.append((int(group), cmd))

    def submit_columns(self, kind, group, from_=None, term=None, id=None, aux=None, flag=None,
                       blk_id=None, blk_next=None) -> None:
        """Queue a whole SoA batch (numpy columns) — jg_submit as is."""
        self._flush_pending()
        n = len(kind)

        def col(x, dt):
            if x is None:
                return None  # an absent column: all zeros, and not even uploaded by jg_step_node
            return np.ascontiguousarray(x, dtype=dt)

        cols = [col(kind, np.uint8), col(group, np.uint32), col(from_, np.uint32), col(term, np.uint64),
                col(id, np.uint64), col(aux, np.uint64), col(flag, np.uint8)]
        bi = np.ascontiguousarray(blk_id if blk_id is not None else [], dtype=np.uint64)
        bn = np.ascontiguousarray(blk_next if blk_next is not None else [], dtype=np.uint64)
        b = capi.CmdBatch()
        b.n = n
        (b.kind, b.group, b.from_, b.term, b.id, b.aux, b.flag) = [None if c is None else c.ctypes.data for c in cols]
        b.n_blocks = len(bi)
        b.blk_id, b.blk_next = bi.ctypes.data, bn.ctypes.data
        self._check(self.api.submit(self._h, C.byref(b)))

    def _flush_pending(self) -> None:
        if not self._pending:
            return
        pend, self._pending = self._pending, []
        n = len(pend)
        kind = np.zeros(n, np.uint8)
        group = np.zeros(n, np.uint32)
        from_ = np.zeros(n, np.uint32)
        term = np.zeros(n, np.uint64)
        id_ = np.zeros(n, np.uint64)
        aux = np.zeros(n, np.uint64)
        flag = np.zeros(n, np.uint8)
        bi: List[int] = []
        bn: List[int] = []
        for i, (g, c) in enumerate(pend):
            kind[i], group[i], from_[i], term[i], flag[i] = c.kind, g, c.from_, c.term, c.flag
            if c.kind == capi.CMD_APPEND_ENTRIES:
                id_[i], aux[i] = len(bi), len(c.blocks)
                for (b_id, b_next) in c.blocks:
                    bi.append(b_id)
                    bn.append(b_next)
            else:
                id_[i], aux[i] = c.id, c.aux
        self.submit_columns(kind, group, from_, term, id_, aux, flag, bi, bn)

    def step(self, now_ms: int = 0) -> None:
        """Apply everything submitted, per group in stream order (Apply::apply)."""
        self._flush_pending()
        self._check(self.api.step(self._h, int(now_ms)))

    def apply(self, group: int, cmd: Command, now_ms: int = 0) -> "RaftHandle":
        """RaftHandle::apply for one group: submit + step."""
        self.submit(group, cmd)
        self.step(now_ms)
        return RaftHandle(self, group)

    def apply_all(self, cmd: Command, now_ms: int = 0) -> None:
        """Apply the same command to every group."""
        n = self.G
        self.submit_columns(np.full(n, cmd.kind, np.uint8), np.arange(n, dtype=np.uint32),
                            np.full(n, cmd.from_, np.uint32), np.full(n, cmd.term, np.uint64),
                            np.full(n, cmd.id, np.uint64), np.full(n, cmd.aux, np.uint64),
                            np.full(n, cmd.flag, np.uint8))
        self.step(now_ms)

    def step_dense_acks(self, acks: np.ndarray) -> None:
        """Dense leader tick from a host [R, G] uint64 array (jg_step_dense_acks)."""
        self._flush_pending()
        a = np.ascontiguousarray(acks, dtype=np.uint64)
        assert a.shape == (self.R, self.G), a.shape
        self._check(self.api.step_dense_acks(self._h, a.ctypes.data))

    def step_node(self, now_ms: int = 0, leader: bool = True, follower: bool = True, tick: bool = True, async_: bool = False,
                  between=None, common_ae: bool = False, fsm_fused: bool = False) -> dict:
        """jg_step_node: a node's whole tick from the rows submitted since the last step - rows in the
        mailbox vocabulary through the dense kernels, everything else through the general state
        machine first - then Command::Tick for every partition.  Returns the outbox columns (host
        copies): beat_term / beat_commit [G], ae [R, G] words, answer / hb_commit [G], and the step's
        row / PCIe byte counts.  common_ae (JG_NODE_COMMON_AE): "aec" [G] is the partition's AppendEntries word
        for every addressee, "ae" only comes down (else None) when some partition's words differ (AEC_INDIVIDUAL);
        fsm_fused (JG_NODE_FSM_FUSED): a leader's rows of a step as one FSM_LEADER_STEP row (`expand_fsm_rows`)."""
        self.step_node_begin(now_ms, leader, follower, tick, async_, common_ae, fsm_fused)
        if between is not None:  # (async_: what the caller does while the step runs - submits for the next one, say)
            between()
        return self.node_outbox()

    def step_node_begin(self, now_ms: int = 0, leader: bool = True, follower: bool = True, tick: bool = True, async_: bool = False,
                        common_ae: bool = False, fsm_fused: bool = False, keep: bool = False) -> None:
        """jg_step_node alone (the outbox: `node_outbox`).  keep (JG_NODE_KEEP, with async_): the step keeps its outputs
        until its outbox is viewed - a second such step may be begun before that; `node_outbox` then serves the OLDER
        one and makes its rows the ones the next drains deliver."""
        self._flush_pending()
        flags = (capi.NODE_LEADER_HALF if leader else 0) | (capi.NODE_FOLLOWER_HALF if follower else 0) | \
                (capi.NODE_TICK if tick else 0) | (capi.NODE_ASYNC if async_ else 0) | \
                (capi.NODE_COMMON_AE if common_ae else 0) | (capi.NODE_FSM_FUSED if fsm_fused else 0) | \
                (capi.NODE_KEEP if keep else 0)
        self._check(self.api.step_node(self._h, int(now_ms), flags))

    def node_outbox(self) -> dict:
        """jg_node_outbox_view as host copies (see `step_node`)."""
        o = capi.NodeOutbox()
        self._check(self.api.node_outbox_view(self._h, C.byref(o)))
        G, R = self.G, self.R

        def arr(p, dt, shape):
            if not p:
                return None
            n = int(np.prod(shape))
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(p)
            return np.frombuffer(buf, dtype=dt).reshape(shape).copy()

        beat = arr(o.beat, np.dtype([("term", "<u8"), ("hb_commit", "<u8")]), (G,))
        return {"beat_term": None if beat is None else beat["term"].copy(),
                "beat_commit": None if beat is None else beat["hb_commit"].copy(),
                "ae": arr(o.ae, np.uint64, (R, G)), "aec": arr(o.aec, np.uint64, (G,)), "answer": arr(o.answer, np.uint64, (G,)),
                "hb_commit": arr(o.hb_commit, np.uint64, (G,)), "rows": int(o.rows), "rows_general": int(o.rows_general),
                "bytes_h2d": int(o.bytes_h2d), "bytes_d2h": int(o.bytes_d2h)}

    def node_inbox_columns(self, slot: int, answer, hb_commit=None) -> None:
        """jg_node_inbox_columns: member slot `slot`'s AppendResponse / HeartbeatResponse for every partition
        as ONE column of JG_ANSWER words (and, optionally, the HeartbeatResponse.commit column) for the next
        step_node - what a batched peer ships instead of two rows per partition."""
        pa, ph = C.c_void_p(), C.c_void_p()
        self._check(self.api.node_inbox_columns(self._h, int(slot), C.byref(pa), C.byref(ph) if hb_commit is not None else None))
        G = self.G
        np.frombuffer((C.c_char * (8 * G)).from_address(pa.value), dtype=np.uint64)[:] = np.asarray(answer, dtype=np.uint64)
        if hb_commit is not None:
            np.frombuffer((C.c_char * (8 * G)).from_address(ph.value), dtype=np.uint64)[:] = np.asarray(hb_commit, dtype=np.uint64)

    def upload_rows(self, kind, group, from_=None, term=None, id=None, aux=None, flag=None,
                    blk_id=None, blk_next=None) -> "DeviceRows":
        """Sort a command batch by group (stable) on the host and park it in device memory for
        step_device_rows — how a caller keeps a pre-built trace resident in HBM."""
        n = len(kind)
        order = np.argsort(np.asarray(group, dtype=np.uint32), kind="stable")

        def col(x, dt):
            a = np.zeros(n, dtype=dt) if x is None else np.asarray(x, dtype=dt)
            return np.ascontiguousarray(a[order])

        cols = [col(kind, np.uint8), col(group, np.uint32), col(from_, np.uint32), col(term, np.uint64),
                col(id, np.uint64), col(aux, np.uint64), col(flag, np.uint8)]
        side = [np.ascontiguousarray(x if x is not None else [], dtype=np.uint64) for x in (blk_id, blk_next)]
        return DeviceRows(self, cols, side)

    def upload_u32(self, values) -> "DeviceList":
        """A list of uint32 (group numbers) parked in device memory: jg_dense_cluster_withdraw_appends / _offer_appends."""
        return DeviceList(self, np.ascontiguousarray(values, dtype=np.uint32))

    def step_device_rows(self, rows: "DeviceRows", now_ms: int = 0) -> None:
        """jg_step_device_rows: apply a device-resident, group-sorted batch (no host pass)."""
        self._flush_pending()
        self._check(self.api.step_device_rows(self._h, C.byref(rows.batch), int(now_ms)))

    def step_dense_acks_n(self, acks: np.ndarray) -> None:
        """T consecutive dense ticks from a host [T, R, G] array.  On the device engine this is
        ONE launch (jg_step_dense_acks_device_n: state read and written once); a backend without
        that entry point gets the T ticks one by one — the two are specified to be identical."""
        self._flush_pending()
        a = np.ascontiguousarray(acks, dtype=np.uint64)
        assert a.ndim == 3 and a.shape[1:] == (self.R, self.G), a.shape
        if not hasattr(self.api, "step_dense_acks_device_n"):
            for t in range(a.shape[0]):
                self._check(self.api.step_dense_acks(self._h, a[t].ctypes.data))
            return
        buf = C.c_void_p()
        self._check(self.api.device_alloc(self._h, a.nbytes, C.byref(buf)))
        try:
            self._check(self.api.device_upload(self._h, buf, a.ctypes.data, a.nbytes))
            self._check(self.api.step_dense_acks_device_n(self._h, buf, a.shape[0]))
            self._check(self.api.sync(self._h))
        finally:
            self.api.device_free(self._h, buf)

    # -- dense node tick (host-array convenience over the device-pointer ABI) ---------------
    def _is_device(self) -> bool:
        return hasattr(self.api, "device_alloc")

    def _to_engine(self, arr: Optional[np.ndarray], keep: list):
        """Host array -> pointer the engine accepts (device copy for the HIP engine)."""
        if arr is None:
            return None
        a = np.ascontiguousarray(arr)
        if not self._is_device():
            keep.append(a)
            return a.ctypes.data
        p = C.c_void_p()
        self._check(self.api.device_alloc(self._h, max(a.nbytes, 16), C.byref(p)))
        self._check(self.api.device_upload(self._h, p, a.ctypes.data, a.nbytes))
        keep.append(p)
        return p.value

    def _out_buffer(self, shape, dtype, keep: list):
        host = np.zeros(shape, dtype=dtype)
        if not self._is_device():
            keep.append(host)
            return host, host.ctypes.data
        p = C.c_void_p()
        self._check(self.api.device_alloc(self._h, max(host.nbytes, 16), C.byref(p)))
        keep.append(p)
        return host, p.value

    def _finish(self, outs, keep) -> None:
        if self._is_device():
            for host, ptr in outs:
                self._check(self.api.device_download(self._h, host.ctypes.data, C.c_void_p(ptr), host.nbytes))
            for k in keep:
                if isinstance(k, C.c_void_p):
                    self.api.device_free(self._h, k)

    def step_dense_leader(self, now_ms: int = 0, acks: Optional[np.ndarray] = None,
                          hbr_has: Optional[np.ndarray] = None, hbr_commit: Optional[np.ndarray] = None,
                          tick: bool = True) -> Optional[dict]:
        """Leader half of a node tick (jg_step_dense_leader) over host arrays in column form: the
        AppendResponse heads (own slot: #appends) and HeartbeatResponse codes are packed into the
        inbox's answer words here, the outbox words are unpacked into {term, hb_commit, ae_from,
        ae_n}; returns None when `tick` is false."""
        assert not self._pending
        keep: list = []
        inbox = capi.LeaderInbox()
        if acks is not None or hbr_has is not None:
            a = np.full((self.R, self.G), capi.NO_ACK, np.uint64) if acks is None else \
                np.asarray(acks, np.uint64).reshape(self.R, self.G)
            if acks is None:  # only HeartbeatResponses: nobody appends
                a[self.read("self_slot"), np.arange(self.G)] = 0
            h = np.full((self.R, self.G), capi.HB_NONE, np.uint8) if hbr_has is None else \
                np.asarray(hbr_has, np.uint8).reshape(self.R, self.G)
            c = np.zeros((self.R, self.G), np.uint64) if hbr_commit is None else \
                np.asarray(hbr_commit, np.uint64).reshape(self.R, self.G)
            inbox.answers = self._to_engine(np.ascontiguousarray(capi.pack_answers(a, h)), keep)
            inbox.hbr_commit = self._to_engine(c, keep)
        outs, res = [], None
        outbox_p = None
        if tick:
            outbox = capi.LeaderOutbox()
            beat, outbox.beat = self._out_buffer((self.G, 2), np.uint64, keep)
            ae, outbox.ae = self._out_buffer((self.R, self.G), np.uint64, keep)
            # (row [own slot] of the block is nobody's mail and is not written while the own slot is the same for
            # every group: JG_NO_ACK by the block's owner, include/josefine_gpu.h jg_leader_outbox)
            ae[:] = np.uint64(capi.NO_ACK)
            if self._is_device():
                self._check(self.api.device_upload(self._h, C.c_void_p(outbox.ae), ae.ctypes.data, ae.nbytes))
            outs = [(beat, outbox.beat), (ae, outbox.ae)]
            outbox_p = C.byref(outbox)
        try:
            self._check(self.api.step_dense_leader(self._h, now_ms, C.byref(inbox), outbox_p))
        finally:
            self._finish(outs, keep)
        if tick:
            ae_from, ae_n = capi.unpack_ae(ae)
            res = {"term": np.ascontiguousarray(beat[:, 0]), "hb_commit": np.ascontiguousarray(beat[:, 1]),
                   "ae_from": ae_from, "ae_n": ae_n}
        return res

    def step_dense_follower(self, now_ms: int, term, hb_commit, ae_from, ae_n, leader=None, leader_id: int = 0,
                            tick: bool = True) -> dict:
        """Follower half of a node tick (jg_step_dense_follower) over host arrays in column form
        (packed into the inbox words here); returns the answers unpacked: {ack_head, hb_commit, hb_has}."""
        assert not self._pending
        keep: list = []
        inbox = capi.FollowerInbox()
        inbox.leader = self._to_engine(None if leader is None else np.asarray(leader, np.uint32), keep)
        inbox.leader_id = int(leader_id)
        beat = np.stack([np.asarray(term, np.uint64), np.asarray(hb_commit, np.uint64)], axis=1)
        inbox.beat = self._to_engine(np.ascontiguousarray(beat), keep)
        inbox.ae = self._to_engine(np.ascontiguousarray(capi.pack_ae(ae_from, ae_n)), keep)
        outbox = capi.FollowerOutbox()
        answer, outbox.answer = self._out_buffer((self.G,), np.uint64, keep)
        hbc, outbox.hb_commit = self._out_buffer((self.G,), np.uint64, keep)
        outs = [(answer, outbox.answer), (hbc, outbox.hb_commit)]
        try:
            self._check(self.api.step_dense_follower(self._h, now_ms, C.byref(inbox), C.byref(outbox), 1 if tick else 0))
        finally:
            self._finish(outs, keep)
        ack_head, hb_has = capi.unpack_answers(answer)
        return {"ack_head": ack_head, "hb_commit": np.where(hb_has != capi.HB_NONE, hbc, np.uint64(0)), "hb_has": hb_has}

    # -- output --------------------------------------------------------------
    def _drain(self, fn, dtype) -> np.ndarray:
        n = C.c_size_t(0)
        self._check(fn(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=dtype)
        if n.value:
            self._check(fn(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def _drain_view(self, fn, dtype, copy: bool) -> np.ndarray:
        """One call: the rows sit in the engine's pinned host queue; `copy=False` returns a
        numpy view of it that is valid until the engine's next synchronising call."""
        p, n = C.c_void_p(), C.c_size_t(0)
        self._check(fn(self._h, C.byref(p), C.byref(n)))
        dt = np.dtype(dtype)
        if not n.value:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n.value * dt.itemsize)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt)
        return arr.copy() if copy else arr

    def drain_messages(self, copy: bool = True) -> np.ndarray:
        """Everything the groups pushed on rpc_tx since the last drain."""
        if hasattr(self.api, "drain_messages_view"):
            return self._drain_view(self.api.drain_messages_view, capi.MSG_DTYPE, copy)
        return self._drain(self.api.drain_messages, capi.MSG_DTYPE)

    def drain_applies(self, copy: bool = True) -> np.ndarray:
        """Everything the groups pushed on fsm_tx since the last drain."""
        if hasattr(self.api, "drain_applies_view"):
            return self._drain_view(self.api.drain_applies_view, capi.FSM_DTYPE, copy)
        return self._drain(self.api.drain_applies, capi.FSM_DTYPE)

    def drain_prefetch(self) -> None:
        """jg_drain_prefetch: start moving everything stepped so far to the host queues without
        blocking (a no-op while the previous batch is still in transfer); from now on the drains
        deliver what has landed and never block."""
        self._flush_pending()
        self._check(self.api.drain_prefetch(self._h))

    def drain_wait(self) -> None:
        """jg_drain_wait: block until the batch in transfer (if any) has landed."""
        self._check(self.api.drain_wait(self._h))

    def drain_flush(self) -> None:
        """jg_drain_flush: block until everything stepped so far has landed in the host queues."""
        self._flush_pending()
        self._check(self.api.drain_flush(self._h))

    def drain_faults(self) -> np.ndarray:
        return self._drain(self.api.drain_faults, capi.FAULT_DTYPE)

    def read(self, field_name: str, replica: int = 0, g0: int = 0, n: Optional[int] = None) -> np.ndarray:
        fld = capi.FIELD_NAMES[field_name]
        n = self.G - g0 if n is None else n
        out = np.zeros(n, dtype=capi.FIELD_DTYPES[fld])
        if n:
            self._check(self.api.read_state(self._h, fld, replica, out.ctypes.data, g0, n))
        return out

    def counters(self) -> dict:
        arr = (C.c_uint64 * 4)()
        self._check(self.api.get_counters(self._h, C.byref(arr)))
        return {"commands": arr[0], "decisions": arr[1], "dense_group_steps": arr[2], "launches": arr[3]}

    def snapshot(self) -> dict:
        """All readable columns (used by the parity tests)."""
        snap = {}
        for name in capi.FIELD_NAMES:
            if name == "match":
                snap[name] = np.stack([self.read("match", r) for r in range(self.R)])
            else:
                snap[name] = self.read(name)
        return snap

    def chain_compact(self, trees: Iterable[Tuple[Sequence[Tuple[int, int]], int]]) -> List[np.ndarray]:
        """Batched Chain::compact: trees = [([(id, next), ...], commit), ...] -> removed masks."""
        trees = list(trees)
        off = np.zeros(len(trees) + 1, np.uint64)
        for i, (blocks, _) in enumerate(trees):
            off[i + 1] = off[i] + len(blocks)
        ids = np.array([b[0] for t in trees for b in t[0]], dtype=np.uint64)
        nexts = np.array([b[1] for t in trees for b in t[0]], dtype=np.uint64)
        commits = np.array([t[1] for t in trees], dtype=np.uint64)
        removed = np.zeros(int(off[-1]), np.uint8)
        ids = np.ascontiguousarray(ids)
        nexts = np.ascontiguousarray(nexts)
        self._check(self.api.chain_compact(self._h, len(trees), off.ctypes.data, ids.ctypes.data,
                                           nexts.ctypes.data, commits.ctypes.data, removed.ctypes.data))
        return [removed[int(off[i]):int(off[i + 1])].copy() for i in range(len(trees))]

    def chain_compact_resident(self) -> np.ndarray:
        """jg_chain_compact_resident + jg_drain_compacted: Chain::compact on every healthy group's own
        chain; returns the removed blocks as (group, id) rows, group ascending, ids in walk order."""
        self._flush_pending()
        n = C.c_size_t(0)
        self._check(self.api.chain_compact_resident(self._h, C.byref(n)))
        return self._drain(self.api.drain_compacted, capi.COMPACT_DTYPE)

    def handle(self, group: int) -> "RaftHandle":
        return RaftHandle(self, group)


class Shard:
    """One shard of a multi-device BatchedRaft: a borrowed engine handle (owned by the parent)
    plus the group range it owns.  Quacks enough like a BatchedRaft for the device-pointer calls."""

    def __init__(self, parent: BatchedRaft, info: "capi.ShardInfo"):
        self.parent, self.api = parent, parent.api
        self._h = C.c_void_p(info.engine)
        self.device_id, self.group_lo, self.G, self.R = info.device_id, info.group_lo, info.n_groups, parent.R
        self.node_ids = parent.node_ids
        self._check = parent._check

    def alloc(self, nbytes: int) -> C.c_void_p:
        p = C.c_void_p()
        self._check(self.api.device_alloc(self._h, nbytes, C.byref(p)))
        return p

    def upload(self, dev_ptr, host: np.ndarray) -> None:
        host = np.ascontiguousarray(host)
        self._check(self.api.device_upload(self._h, dev_ptr, host.ctypes.data, host.nbytes))


class RaftHandle:
    """enum RaftHandle (src/raft/mod.rs:417-468) for one group of a BatchedRaft."""

    def __init__(self, engine: BatchedRaft, group: int):
        self.engine = engine
        self.group = int(group)

    def apply(self, cmd: Command, now_ms: int = 0) -> "RaftHandle":
        return self.engine.apply(self.group, cmd, now_ms)

    def _get(self, name: str, replica: int = 0):
        return self.engine.read(name, replica, self.group, 1)[0]

    def is_follower(self) -> bool:
        return self._get("role") == capi.ROLE_FOLLOWER

    def is_candidate(self) -> bool:
        return self._get("role") == capi.ROLE_CANDIDATE

    def is_leader(self) -> bool:
        return self._get("role") == capi.ROLE_LEADER

    @property
    def id(self) -> int:
        return self.engine.node_ids[int(self._get("self_slot"))]

    @property
    def current_term(self) -> int:
        return int(self._get("term"))

    @property
    def voted_for(self) -> Optional[int]:
        return int(self._get("voted_for")) if self._get("has_voted") else None

    @property
    def commit(self) -> int:
        return int(self._get("commit"))

    @property
    def head(self) -> int:
        return int(self._get("head"))

    @property
    def fault(self) -> int:
        return int(self._get("fault"))

    def match(self, replica: int) -> int:
        return int(self._get("match", replica))


class DeviceRows:
    """A command batch resident in device memory (see BatchedRaft.upload_rows)."""

    def __init__(self, engine: BatchedRaft, cols, side):
        self.engine = engine
        self.n = len(cols[0])
        self._ptrs = []
        api, h = engine.api, engine._h

        def up(a):
            p = C.c_void_p()
            engine._check(api.device_alloc(h, max(a.nbytes, 16), C.byref(p)))
            if a.nbytes:
                engine._check(api.device_upload(h, p, a.ctypes.data, a.nbytes))
            self._ptrs.append(p)
            return p.value

        b = capi.CmdBatch()
        b.n = self.n
        (b.kind, b.group, b.from_, b.term, b.id, b.aux, b.flag) = [up(c) for c in cols]
        b.n_blocks = len(side[0])
        b.blk_id, b.blk_next = up(side[0]), up(side[1])
        self.batch = b

    def free(self) -> None:
        for p in self._ptrs:
            self.engine.api.device_free(self.engine._h, p)
        self._ptrs = []


class DeviceList:
    """uint32 values in device memory (jg_device_alloc + jg_device_upload); .ptr is the device address."""

    def __init__(self, engine, values: np.ndarray):
        self.engine, self.n = engine, len(values)
        p = C.c_void_p()
        engine._check(engine.api.device_alloc(engine._h, max(values.nbytes, 16), C.byref(p)))
        self._p = p
        if values.nbytes:
            engine._check(engine.api.device_upload(engine._h, p, values.ctypes.data, values.nbytes))

    @property
    def ptr(self) -> int:
        return self._p.value

    def free(self) -> None:
        if self._p:
            self.engine.api.device_free(self.engine._h, self._p)
            self._p = None


class DenseCluster:
    """jg_dense_cluster: the R nodes of every partition as R engines of one process, their
    protocol rounds driven from inside the library (josefine_gpu.h, "a closed loop of dense node
    ticks").  `nodes[r]` hosts replica slot r of every group; nodes[lead] is made leader by the
    caller (traces.elect_all)."""

    def __init__(self, nodes, lead=0, vote_words=False):
        """lead = None: per-partition leadership (JG_CLUSTER_ANY_LEADER) - every node leads the partitions it was
        elected for and follows the others.  vote_words: JG_CLUSTER_OPT_VOTE_WORDS (an election's traffic of the routed
        round as mailbox words, csrc/jg_votes.h)."""
        self.nodes, self.lead, self.R = list(nodes), lead, len(nodes)
        self.api = nodes[0].api
        arr = (C.c_void_p * self.R)(*[n._h for n in nodes])
        self._h = C.c_void_p()
        nodes[0]._check(self.api.dense_cluster_create(arr, self.R, capi.CLUSTER_ANY_LEADER if lead is None else lead, C.byref(self._h)))
        self.vote_words = bool(vote_words)
        if vote_words:
            self.set_option(capi.CLUSTER_OPT_VOTE_WORDS, 1)

    def set_option(self, option: int, value: int) -> None:
        self.nodes[0]._check(self.api.dense_cluster_set_option(self._h, int(option), int(value)))

    def close(self) -> None:
        if self._h:
            self.api.dense_cluster_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # (this object keeps its nodes alive: the cluster always goes before them)
        try:
            self.close()
        except Exception:
            pass

    def set_appends(self, uniform: int = 1, per_group=None) -> None:
        p = None if per_group is None else np.ascontiguousarray(per_group, dtype=np.uint64).ctypes.data
        self.nodes[0]._check(self.api.dense_cluster_set_appends(self._h, int(uniform), p))

    def withdraw_appends(self, groups_dev_ptr: int, n: int) -> None:
        """jg_dense_cluster_withdraw_appends: no more ClientRequests for the n groups listed in device memory."""
        self.nodes[0]._check(self.api.dense_cluster_withdraw_appends(self._h, C.c_void_p(groups_dev_ptr), int(n)))

    def offer_appends(self, groups_dev_ptr: int, n: int, per_round: int) -> None:
        """jg_dense_cluster_offer_appends: `per_round` ClientRequests per round for the n groups listed in device memory."""
        self.nodes[0]._check(self.api.dense_cluster_offer_appends(self._h, C.c_void_p(groups_dev_ptr), int(n), int(per_round)))

    def rounds(self, now_ms: int, dt_ms: int, n: int) -> None:
        self.nodes[0]._check(self.api.dense_cluster_rounds(self._h, int(now_ms), int(dt_ms), int(n)))

    def round_routed(self, now_ms: int, inject=None) -> dict:
        """jg_dense_cluster_round_routed; `inject`: per node a DeviceRows (upload_rows) or None."""
        arr = None
        if inject is not None:
            arr = (capi.CmdBatch * self.R)()
            for r, rows in enumerate(inject):
                if rows is not None:
                    arr[r] = rows.batch
        st = capi.RouteStats()
        self.nodes[0]._check(self.api.dense_cluster_round_routed(self._h, int(now_ms), arr, C.byref(st)))
        return {"delivered": [int(st.delivered[r]) for r in range(self.R)], "kept": int(st.kept), "fsm_rows": int(st.fsm_rows)}
