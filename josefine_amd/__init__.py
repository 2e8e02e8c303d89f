"""josefine_amd — MI355X-native batched Chained-Raft engine (hot path of
tychedelia/josefine's src/raft behind a C ABI)."""
from . import _capi as capi
from .engine import BatchedRaft, Command, DenseCluster, DeviceRows, EngineError, RaftHandle, device_api, expand_fsm_rows

__all__ = ["BatchedRaft", "Command", "DenseCluster", "DeviceRows", "EngineError", "RaftHandle", "capi", "device_api", "expand_fsm_rows"]
