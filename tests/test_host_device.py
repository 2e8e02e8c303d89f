"""The engine's own translation unit on an EMULATED device (CPU; tests/host_device.py): josefine_gpu.hip - the C ABI, its
host code and every kernel as written - compiled for the host against a stand-in for the HIP runtime in which a workgroup
is 256 cooperative fibers, loaded by a CHILD process through JOSEFINE_GPU_LIB, and driven by the GPU suite's own tests.

What runs here is what no GPU-minute was left for at the end of round 4: the routed round with the election vocabulary
as mailbox words (JG_CLUSTER_OPT_VOTE_WORDS, josefine_amd/csrc/jg_votes.h) THROUGH round_routed_impl - its job tables, step
numbers, the mail's two buffers, the census before and the word-aware delivering pass inside the repeat loop, the
exceptional queues' clean-up - against the oracle clusters that move every message as a row (tests/test_gpu_vote_words.py,
the cases small enough for fibers).  And, so that the stand-in itself is held to something, a slice of the default path's
GPU tests, which the real device passes."""
import pytest

import host_device


def _run(args, env=None):
    r = host_device.run_pytest(args, env=dict(JG_NO_GRAPH="1", **(env or {})))
    tail = r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert r.returncode == 0, tail
    return r.stdout


def test_routed_round_with_the_vote_mail_through_the_engines_host_code():
    """single lead (BASELINE configs[4]'s cluster) and per-partition leadership, under the switch"""
    # (a minute of fibers; every case whose id starts with "small" runs here in five: -k small)
    out = _run(["tests/test_gpu_vote_words.py", "-m", "gpu", "-k", "small-5-25-3-160 or small-3-25-7-160"])
    assert "2 passed" in out, out[-500:]


def test_the_default_path_on_the_emulated_device():
    """the stand-in held to tests the real device passes: the sparse step under the fuzzed command streams (every role, the
    chain, elections), the dense tick against the sparse path, the T-tick kernel, device-resident rows - through the C ABI"""
    out = _run(["tests/test_gpu_parity.py", "-m", "gpu", "-k", "test_fuzz_command_stream_parity or test_fuzz_state_aware_stream_parity or "
                "test_dense_equals_sparse_path or test_dense_fused_ticks_parity or test_device_resident_rows_path or test_election_setup_parity"])
    assert "22 passed" in out, out[-500:]


def test_the_node_steps_bus_formats_on_the_emulated_device():
    """ABI v7's formats of the node step - the packed kind byte, the common AppendEntries word, the fused fsm row - through
    the engine's host code (early uploads, the asynchronous step's settling, the on-demand fetch of the rows) against the
    oracle's step over the plain rows"""
    out = _run(["tests/test_node_step.py", "-m", "gpu", "-k", "(compact_bus and (600 or 2000)) or outlive"])
    assert "3 passed" in out, out[-500:]


def test_smoke_on_the_emulated_device():
    """__graft_entry__.smoke() as the driver calls it on the MI355X, here against the emulated build: elections through the
    general kernel, eight dense ticks of the ragged stream, six ticks through the Apply surface - every column, row and
    word against the oracle (the two seconds that say the entry point itself is not what breaks on the GPU box)"""
    import os
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(JOSEFINE_GPU_LIB=host_device.build(), JG_EMULATED_DEVICE="1", JG_NO_GRAPH="1")
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], env=e, capture_output=True, text=True, cwd=host_device.ROOT, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_the_product_refuses_the_emulated_build():
    """the emulated build is the tests': pointed at it without saying so (JG_EMULATED_DEVICE=1, which only the tests' child
    processes set), josefine_amd fails loudly - there is no way to run the product without the gfx950 library"""
    import os
    import subprocess
    import sys
    e = dict(os.environ)
    e.pop("JG_EMULATED_DEVICE", None)
    e.update(JOSEFINE_GPU_LIB=host_device.build())
    r = subprocess.run([sys.executable, "-c", "from josefine_amd import BatchedRaft; BatchedRaft(4, 1)"], env=e, capture_output=True, text=True, cwd=host_device.ROOT,
                       timeout=300)
    assert r.returncode != 0 and "emulated-device build" in r.stderr and "no CPU fallback" in r.stderr, r.stderr[-1500:]
