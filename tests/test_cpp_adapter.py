"""Build and run the C++ host-adapter tests (tests/cpp/test_raft_handle.cpp) against
the HIP engine: the reference's L1 tests through josefine_amd/host/raft_handle.hpp and
BASELINE config #1 (3-node plumbing)."""
import os
import subprocess

import pytest

from josefine_amd.build import CSRC, build_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_raft_handle.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_raft_handle")


def compile_adapter_test():
    build_hip()
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-o", EXE, SRC, f"-L{CSRC}", "-ljosefine_gpu",
                    f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib"], check=True)


def test_cpp_adapter_compiles():
    """CPU: the header-only adapter and its tests compile and link against the C ABI."""
    compile_adapter_test()
    assert os.path.exists(EXE)


def test_cpp_host_logic_on_oracle():
    """CPU: the host logic above the ABI (BatchedRaft pump, BlockStore, fsm::Driver and
    server::event_loop for many partitions) with the oracle library standing in for the engine:
    the reference's L1 tests, its event_loop test and the 3-node plumbing of BASELINE config #1."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-C", odir], check=True, capture_output=True)
    exe = EXE + "_oracle"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-DJG_TEST_AGAINST_ORACLE", "-o", exe, SRC, f"-L{odir}",
                    "-ljosefine_oracle", f"-Wl,-rpath,{odir}"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpp adapter ok" in r.stdout


def test_host_formats_kats():
    """CPU: josefine_amd/host/formats.hpp — the sled key / bincode value encoding of the chain store
    and the length-delimited serde_json peer protocol, against the
    reference's own vectors (chain.rs:345-350, tcp.rs:172-232) and one vector per Command."""
    src = os.path.join(ROOT, "tests", "cpp", "test_formats.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "test_formats")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-o", exe, src], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "formats ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_adapter_runs_reference_tests():
    compile_adapter_test()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpp adapter ok" in r.stdout
