"""In-tree build of the HIP engine: one hipcc invocation, gfx950 only."""
from __future__ import annotations

import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libjosefine_gpu.so")
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".hip")))  # every header the one translation unit includes
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, s) for s in SOURCES]
    deps.append(os.path.join(CSRC, "..", "..", "include", "josefine_gpu.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> josefine_amd/csrc/libjosefine_gpu.so"""
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wall",
           "-Wno-unused-function", "-o", LIB, os.path.join(CSRC, "josefine_gpu.hip")]
    if os.environ.get("JG_BLOCK"):  # workgroup-size experiments (profiles/README.md); default 256
        cmd.insert(1, "-DJG_BLOCK=" + os.environ["JG_BLOCK"])
    for k in ("JG_LEADER_WAVES", "JG_FOLLOWER_WAVES"):  # occupancy experiments on the dense halves
        if os.environ.get(k):
            cmd.insert(1, f"-D{k}=" + os.environ[k])
    if os.environ.get("JG_GSM_WAVES"):  # occupancy experiment: the general state machine's kernels held to 512 / N VGPRs
        cmd.insert(1, "-DJG_GSM_WAVES=" + os.environ["JG_GSM_WAVES"])
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
EVENT_LOOP_BENCH = os.path.join(HOST, "bench_event_loop")


def build_event_loop_bench(force: bool = False) -> str:
    """g++ -> josefine_amd/host/bench_event_loop: josefine::BatchedEventLoop (host/raft_handle.hpp) over
    libjosefine_gpu.so, the binary behind `bench.py --event-loop`."""
    src = os.path.join(HOST, "bench_event_loop.cpp")
    deps = [src, os.path.join(HOST, "raft_handle.hpp"), os.path.join(CSRC, "..", "..", "include", "josefine_gpu.h")]
    build_hip()
    if not force and os.path.exists(EVENT_LOOP_BENCH) and all(os.path.getmtime(d) <= os.path.getmtime(EVENT_LOOP_BENCH) for d in deps):
        return EVENT_LOOP_BENCH
    subprocess.run(["g++", "-std=c++17", "-O2", "-march=native", "-Wall", "-o", EVENT_LOOP_BENCH, src, f"-L{CSRC}", "-ljosefine_gpu",
                    f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EVENT_LOOP_BENCH
