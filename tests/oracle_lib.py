"""Test-side loader for the CPU oracle (oracle/libjosefine_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch
oracle/.  The oracle exports the ABI of include/josefine_gpu.h under the `jo_`
prefix, so `BatchedRaft(api=load_oracle())` drives it with the same host code
as the HIP engine.
"""
import os
import subprocess

from josefine_amd import BatchedRaft, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libjosefine_oracle.so")
_api = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load_oracle() -> capi.Api:
    global _api
    if _api is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle_engine.cpp", "raft_oracle.hpp")]
        srcs.append(os.path.join(ROOT, "include", "josefine_gpu.h"))
        if (not os.path.exists(ORACLE_SO)
                or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs)):
            build_oracle()
        _api = capi.Api(ORACLE_SO, "jo_")
    return _api


def oracle_engine(*args, **kw) -> BatchedRaft:
    return BatchedRaft(*args, api=load_oracle(), **kw)
