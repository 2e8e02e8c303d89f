"""BASELINE.json configs[4] trace helpers — the generators live in josefine_amd.traces so that
bench.py and the tests build identical rows; what happens to those rows is the reference's
truth: a re-elected leader that had committed anything trips assert!(id > head) on its first
append (chain.rs:163, Q8), so failed groups drop out until they are restarted again."""
from josefine_amd.traces import failure_rows, mix64, synth_hash  # noqa: F401
