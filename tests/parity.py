"""Shared helpers for the parity tests: drive the HIP engine and the CPU oracle
with identical inputs and compare every readable state column and every
drained output row bit for bit."""
import ctypes as C

import numpy as np

from josefine_amd import BatchedRaft, Command, capi


def compare_snapshots(dev: BatchedRaft, ora: BatchedRaft, what: str = "", fields=None) -> None:
    names = list(fields or capi.FIELD_NAMES)
    if "fault" in names:  # report a fault mismatch before its consequences
        names.remove("fault")
        names.insert(0, "fault")
    for name in names:
        if name == "match":
            for r in range(dev.R):
                a, b = dev.read("match", r), ora.read("match", r)
                if not np.array_equal(a, b):
                    bad = np.nonzero(a != b)[0]
                    raise AssertionError(f"{what}: match[{r}] differs at groups {bad[:8]}: "
                                         f"hip={a[bad[:8]]} oracle={b[bad[:8]]}")
        else:
            a, b = dev.read(name), ora.read(name)
            if not np.array_equal(a, b):
                bad = np.nonzero(a != b)[0]
                raise AssertionError(f"{what}: {name} differs at groups {bad[:8]}: "
                                     f"hip={a[bad[:8]]} oracle={b[bad[:8]]}")


def compare_drains(dev: BatchedRaft, ora: BatchedRaft, what: str = "") -> None:
    for fn in ("drain_messages", "drain_applies", "drain_faults"):
        a, b = getattr(dev, fn)(), getattr(ora, fn)()
        if a.shape != b.shape or a.tobytes() != b.tobytes():
            n = min(len(a), len(b))
            first = next((i for i in range(n) if a[i].tobytes() != b[i].tobytes()), n)
            raise AssertionError(f"{what}: {fn} differs (hip {len(a)} rows, oracle {len(b)} rows) at row {first}: "
                                 f"hip={a[first] if first < len(a) else None} "
                                 f"oracle={b[first] if first < len(b) else None}")


from josefine_amd.traces import elect_all  # noqa: E402,F401  (shared with bench.py / smoke)


def synth_tick_host(ora: BatchedRaft, mode: int, tick: int, sim: np.ndarray) -> np.ndarray:
    """The oracle-side restatement of the synthetic ack generator (host arrays)."""
    acks = np.zeros((ora.R, ora.G), dtype=np.uint64)
    rc = ora.api.synth_fill_acks(ora._h, mode, tick, sim.ctypes.data, acks.ctypes.data)
    assert rc == 0, ora.api.error()
    return acks


class DeviceSynth:
    """Device-resident generator state + one ack buffer."""

    def __init__(self, dev: BatchedRaft):
        self.dev = dev
        self.bytes = dev.R * dev.G * 8
        self.sim = C.c_void_p()
        self.acks = C.c_void_p()
        dev._check(dev.api.device_alloc(dev._h, self.bytes, C.byref(self.sim)))
        dev._check(dev.api.device_alloc(dev._h, self.bytes, C.byref(self.acks)))

    def fill(self, mode: int, tick: int) -> None:
        d = self.dev
        d._check(d.api.synth_fill_acks_device(d._h, mode, tick, self.sim, self.acks))

    def download_acks(self) -> np.ndarray:
        out = np.zeros((self.dev.R, self.dev.G), dtype=np.uint64)
        self.dev._check(self.dev.api.device_download(self.dev._h, out.ctypes.data, self.acks, self.bytes))
        return out

    def close(self) -> None:
        self.dev.api.device_free(self.dev._h, self.sim)
        self.dev.api.device_free(self.dev._h, self.acks)


def run_dense_ticks(dev: BatchedRaft, ora: BatchedRaft, mode: int, ticks: int, check_every: int = 1,
                    fields=("commit", "head", "match", "repl_state", "fault", "id_gen", "role", "term")) -> None:
    """Generate the ack stream on the device, check the generator against its host
    restatement, apply the tick on both sides, compare state."""
    synth = DeviceSynth(dev)
    sim = np.zeros((ora.R, ora.G), dtype=np.uint64)
    try:
        for t in range(ticks):
            synth.fill(mode, t)
            host_acks = synth_tick_host(ora, mode, t, sim)
            if t % check_every == 0 or t == ticks - 1:
                assert np.array_equal(synth.download_acks(), host_acks), f"tick {t}: generators disagree"
            dev._check(dev.api.step_dense_acks_device(dev._h, synth.acks))
            ora.step_dense_acks(host_acks)
            if t % check_every == 0 or t == ticks - 1:
                compare_snapshots(dev, ora, f"dense tick {t}", fields)
                compare_drains(dev, ora, f"dense tick {t}")
    finally:
        synth.close()
    assert dev.counters()["decisions"] == ora.counters()["decisions"], (dev.counters(), ora.counters())
