"""CPU-side checks of the boundary: the C-ABI library builds, loads and exports every
symbol include/josefine_gpu.h declares; without a GPU it refuses to run (there is
no CPU fallback); the host mirror encodes Commands as the header specifies."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from josefine_amd import BatchedRaft, Command, EngineError, capi
from josefine_amd.build import LIB, build_hip
from oracle_lib import oracle_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "josefine_gpu.h")


@pytest.fixture(scope="module")
def lib():
    build_hip()  # hipcc cross-compiles gfx950 without a GPU
    return C.CDLL(LIB)


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jg_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_match_binding_table():
    assert declared_symbols() == sorted(capi.HEADER_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.jg_abi_version.restype = C.c_uint32
    assert lib.jg_abi_version() == capi.ABI_VERSION


def test_library_contains_gfx950_code_object():
    out = os.popen(f"/opt/rocm/lib/llvm/bin/llvm-readelf --notes {LIB} 2>/dev/null; strings {LIB} | grep -m1 gfx950").read()
    assert "gfx950" in out


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product path must fail loudly, not compute on the host."""
    api = capi.Api(LIB, "jg_")
    cfg = capi.Config()
    cfg.abi_version, cfg.n_groups, cfg.n_replicas = capi.ABI_VERSION, 4, 1
    cfg.node_ids[0] = 1
    cfg.heartbeat_timeout_ms, cfg.election_timeout_min_ms, cfg.election_timeout_max_ms = 100, 500, 1000
    h = C.c_void_p()
    rc = api.engine_create(C.byref(cfg), C.byref(h))
    if rc == capi.OK:  # a GPU is present (GPU box): nothing to check here
        api.engine_destroy(h)
        pytest.skip("GPU present")
    assert rc == capi.EDEVICE
    assert "no CPU fallback" in api.error() or "HIP" in api.error() or "hip" in api.error()


def test_package_never_loads_the_oracle():
    import josefine_amd
    pkg = os.path.dirname(josefine_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "raft_oracle" not in text and "libjosefine_oracle" not in text, f
                assert "oracle_engine" not in text, f


def test_row_struct_layouts_match_header():
    assert np.dtype(capi.MSG_DTYPE).itemsize == 40
    assert np.dtype(capi.FSM_DTYPE).itemsize == 24
    assert np.dtype(capi.FAULT_DTYPE).itemsize == 8
    assert C.sizeof(capi.Config) == 160  # ABI v2: the v1 88 bytes + n_devices + device_ids[16] (+ 4 B tail padding)


def test_ctypes_structs_match_the_c_header(tmp_path):
    """sizeof / offsetof of every struct of include/josefine_gpu.h as gcc lays them out, against
    the ctypes mirrors in josefine_amd/_capi.py."""
    import subprocess
    structs = {"jg_config": capi.Config, "jg_cmd_batch": capi.CmdBatch, "jg_shard_info": capi.ShardInfo,
               "jg_leader_inbox": capi.LeaderInbox, "jg_leader_outbox": capi.LeaderOutbox,
               "jg_follower_inbox": capi.FollowerInbox, "jg_follower_outbox": capi.FollowerOutbox,
               "jg_route_stats": capi.RouteStats}
    rename = {"from_": "from"}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "josefine_gpu.h")}"',
             'int main(void) {']
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, *_ in ct._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {rename.get(fname, fname)}));')
    lines += ['  printf("jg_msg_row %zu\\n", sizeof(jg_msg_row));', '  printf("jg_fsm_row %zu\\n", sizeof(jg_fsm_row));',
              '  printf("jg_leader_beat %zu\\n", sizeof(jg_leader_beat));',
              '  printf("answer %llx\\n", (unsigned long long)JG_ANSWER(5, JG_HB_NONE));',
              '  printf("ae %llx\\n", (unsigned long long)JG_AE(7, 2));',
              '  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, ct in structs.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        for fname, *_ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, (cname, fname)
    assert int(got["jg_msg_row"]) == np.dtype(capi.MSG_DTYPE).itemsize
    assert int(got["jg_fsm_row"]) == np.dtype(capi.FSM_DTYPE).itemsize
    # the mailbox words: a 16-byte beat, and the header's packing macros against the Python helpers
    assert int(got["jg_leader_beat"]) == 16
    assert int(got["answer"], 16) == int(capi.pack_answers(np.array([5], np.uint64), np.array([capi.HB_NONE], np.uint8))[0])
    assert int(got["ae"], 16) == int(capi.pack_ae(np.array([7], np.uint64), np.array([2], np.uint8))[0])
    assert capi.unpack_answers(capi.pack_answers(np.array([capi.NO_ACK, 9], np.uint64), np.array([1, capi.HB_NONE], np.uint8)))[0].tolist() == [capi.NO_ACK, 9]


def test_command_constructors_use_header_columns():
    c = Command.VoteRequest(term=7, candidate_id=3, last_term=6, head=9)
    assert (c.kind, c.from_, c.term, c.id, c.aux) == (capi.CMD_VOTE_REQUEST, 3, 7, 9, 6)
    c = Command.AppendResponse(node_id=2, term=1, head=5)
    assert (c.kind, c.from_, c.id, c.flag) == (capi.CMD_APPEND_RESPONSE, 2, 5, 1)
    c = Command.Heartbeat(term=4, commit=3, leader_id=2)
    assert (c.kind, c.from_, c.term, c.id) == (capi.CMD_HEARTBEAT, 2, 4, 3)
    c = Command.AppendEntries(1, 2, [(1, 0), (2, 1)])
    assert c.blocks == [(1, 0), (2, 1)]


def test_stream_order_across_submits_and_groups():
    """Rows of one group apply in submit order, whatever the interleaving with other groups."""
    e = oracle_engine(3, 3)
    e.submit(2, Command.Timeout())
    e.submit(0, Command.Timeout())
    e.submit(2, Command.VoteResponse(1, 2, True))   # elected
    e.submit(2, Command.ClientRequest(5))
    e.submit(0, Command.VoteResponse(1, 2, False))
    e.submit(2, Command.AppendResponse(3, 1, 1))    # commit 1
    e.submit(0, Command.VoteResponse(1, 3, False))  # defeated
    e.step()
    assert list(e.read("role")) == [capi.ROLE_FOLLOWER, capi.ROLE_FOLLOWER, capi.ROLE_LEADER]
    assert list(e.read("commit")) == [0, 0, 1]
    m = e.drain_messages()
    assert list(m["group"]) == sorted(m["group"])  # drained group-major, in emission order per group
    assert [int(k) for k in m["kind"][m["group"] == 2]] == [capi.CMD_VOTE_REQUEST] * 2 + [capi.CMD_HEARTBEAT]


def test_drain_capacity_contract():
    e = oracle_engine(1, 3)
    e.apply(0, Command.Timeout())
    n = C.c_size_t(0)
    assert e.api.drain_messages(e._h, None, 0, C.byref(n)) == capi.OK and n.value == 2
    buf = np.zeros(1, dtype=capi.MSG_DTYPE)
    assert e.api.drain_messages(e._h, buf.ctypes.data, 1, C.byref(n)) == capi.ECAPACITY
    assert len(e.drain_messages()) == 2  # nothing was consumed by the failed call


def test_submit_validation():
    e = oracle_engine(2, 3)
    with pytest.raises(EngineError):
        e.submit_columns([capi.CMD_TICK], [5])            # group out of range
    with pytest.raises(EngineError):
        e.submit_columns([99], [0])                       # unknown kind
    with pytest.raises(EngineError):
        e.submit_columns([capi.CMD_APPEND_ENTRIES], [0], id=[0], aux=[3], blk_id=[1], blk_next=[0])


def test_header_is_plain_c():
    """The boundary is a C ABI: include/josefine_gpu.h must compile as C99 (what cgo / bindgen /
    a Rust `extern "C"` block would be generated from), with no C++ or HIP types in it."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "josefine_gpu.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = open(hdr).read()
    assert "hip" not in src.lower().replace("hipcc", "") or "hip_runtime" not in src
    assert "torch" not in src.lower()


def test_integration_md_binds_every_entry_point():
    """INTEGRATION.md's `extern "C"` block (the binding a josefine maintainer would paste into
    src/raft/gpu/ffi.rs) names exactly the functions include/josefine_gpu.h declares."""
    import re
    hdr = open(os.path.join(ROOT, "include", "josefine_gpu.h")).read()
    declared = set(re.findall(r"^(?:int|void|uint32_t|const char\*) (jg_[a-z_0-9]+)\(", hdr, re.M))
    bound = set(re.findall(r"pub fn (jg_[a-z_0-9]+)\(", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    assert declared == set(capi.HEADER_SYMBOLS)


# ---- the documented Rust binding against the header, value by value and byte by byte ----------------
def _rust_block():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"## 3\. `src/raft/gpu/ffi\.rs`.*?```rust\n(.*?)```", text, re.S)
    assert m, "INTEGRATION.md §3 has no rust block"
    return re.sub(r"//[^\n]*", "", m.group(1))


_RUST_PRIM = {"u8": (1, 1), "i8": (1, 1), "u16": (2, 2), "i16": (2, 2), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4),
              "u64": (8, 8), "i64": (8, 8), "usize": (8, 8), "isize": (8, 8), "c_int": (4, 4)}


def _rust_consts(block):
    out = {}
    for name, ty, val in re.findall(r"pub const (\w+): (\w+) = ([^;]+);", block):
        v = val.replace("u64::MAX", str(2**64 - 1)).strip()
        out[name] = (ty, int(eval(v, {"__builtins__": {}}, dict((k, x[1]) for k, x in out.items()))))  # noqa: S307 (our own document)
    return out


def _rust_layout(ty, consts, structs):
    """(size, align) of a Rust type under #[repr(C)] on x86-64."""
    ty = ty.strip()
    if ty.startswith("*const ") or ty.startswith("*mut "):
        return 8, 8
    m = re.fullmatch(r"\[(.+);\s*(\w+)\]", ty)
    if m:
        n = int(m.group(2)) if m.group(2).isdigit() else consts[m.group(2)][1]
        s, a = _rust_layout(m.group(1), consts, structs)
        return s * n, a
    if ty in _RUST_PRIM:
        return _RUST_PRIM[ty]
    fields = structs[ty]
    off, align = 0, 1
    for _, fty in fields:
        s, a = _rust_layout(fty, consts, structs)
        off = (off + a - 1) // a * a + s
        align = max(align, a)
    return (off + align - 1) // align * align, align


def _rust_structs(block):
    out = {}
    for name, body in re.findall(r"pub struct (\w+)\s*\{([^}]*)\}", block):
        fields = re.findall(r"pub (\w+):\s*((?:\[[^\]]*\]|[^,\[])+?)\s*(?:,|$)", body.strip())
        out[name] = [(f, t.strip()) for f, t in fields]
    return out


def _header_structs():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef struct \w+\s*\{(.*?)\}\s*(\w+);", text, re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):  # `uint64_t a, b` declares two fields
                fields.append(re.search(r"(\w+)\s*(?:\[[^\]]*\])?\s*$", part.strip()).group(1))
        out[name] = fields
    return out


def _c_facts(struct_fields, const_names):
    """sizeof / offsetof / constant values as the C compiler sees include/josefine_gpu.h."""
    import json
    import subprocess
    import tempfile
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {", 'printf("{\\n");']
    for s, fields in struct_fields.items():
        lines.append(f'printf("\\"sizeof {s}\\": %zu,\\n", sizeof({s}));')
        for f in fields:
            lines.append(f'printf("\\"{s}.{f}\\": [%zu, %zu],\\n", offsetof({s}, {f}), sizeof((({s}*)0)->{f}));')
    for c in const_names:
        lines.append(f'printf("\\"{c}\\": %lld,\\n", (long long)({c}));')
    lines += ['printf("\\"_\\": 0}\\n");', "return 0; }"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "abi.c"), os.path.join(d, "abi")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-std=c99", "-o", exe, src], check=True)
        return json.loads(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)


def test_integration_md_rust_binding_matches_the_header_byte_for_byte():
    """Every `pub const` of the documented Rust block has the header's value (JG_ABI_VERSION above all:
    jg_engine_create rejects a mismatch), every #[repr(C)] struct has the header's fields in the header's
    order at the header's offsets and sizes, and every `pub fn` has the header's parameter list (pointer /
    integer width per position).  Round 2's document declared version 2 against a version-3 library."""
    block = _rust_block()
    consts, rstructs, cstructs = _rust_consts(block), _rust_structs(block), _header_structs()
    opaque = {n for n, f in rstructs.items() if [x[0] for x in f] == [] or "_private" in block.split(f"pub struct {n}")[1][:40]}
    named = {n: f for n, f in rstructs.items() if n not in opaque}
    assert set(named) == set(cstructs), (sorted(set(named) ^ set(cstructs)))
    c_names = set(re.findall(r"\b(JG_[A-Z_0-9]+)\b", open(HEADER).read()))
    shared = sorted(set(consts) & c_names)
    assert "JG_ABI_VERSION" in shared and len(shared) >= 30
    assert not set(consts) - c_names - {"JG_MAILBOX_NONE"}, sorted(set(consts) - c_names)
    facts = _c_facts(cstructs, shared + ["JG_MAILBOX_NONE"])
    for c in shared + ["JG_MAILBOX_NONE"]:
        want = facts[c] & (2**64 - 1) if consts[c][0] in ("u64", "usize") else facts[c]
        assert consts[c][1] == want, (c, consts[c][1], want)
    assert consts["JG_ABI_VERSION"][1] == capi.ABI_VERSION
    for s, fields in named.items():
        assert [f for f, _ in fields] == cstructs[s], (s, [f for f, _ in fields], cstructs[s])
        off = 0
        for f, ty in fields:
            size, align = _rust_layout(ty, consts, rstructs)
            off = (off + align - 1) // align * align
            assert [off, size] == facts[f"{s}.{f}"], (s, f, ty, [off, size], facts[f"{s}.{f}"])
            off += size
        assert _rust_layout(s, consts, rstructs)[0] == facts[f"sizeof {s}"], s
    # functions: the same number of parameters, pointer vs 4 / 8-byte integer vs float per position
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)

    def c_class(p):
        p = p.strip()
        if "*" in p or "[" in p:  # (an array parameter is a pointer)
            return "ptr"
        t = re.sub(r"\b(const|\w+$)\b", "", p).strip() if " " in p else p
        return {"uint64_t": "i8", "size_t": "i8", "uint32_t": "i4", "int": "i4", "int32_t": "i4", "float": "f4", "void": "void"}[t.strip()]

    def rust_class(p):
        ty = p.split(":", 1)[1].strip()
        if ty.startswith("*"):
            return "ptr"
        return {"u64": "i8", "usize": "i8", "u32": "i4", "c_int": "i4", "i32": "i4", "f32": "f4"}[ty]

    cf = {n: [c_class(a) for a in args.split(",") if a.strip() and a.strip() != "void"]
          for n, args in re.findall(r"^(?:int|void|uint32_t|const char\*) (jg_\w+)\(([^)]*)\);", hdr, re.M | re.S)}
    rf = {n: [rust_class(a) for a in re.split(r",(?![^\[]*\])", args) if a.strip()]
          for n, args in re.findall(r"pub fn (jg_\w+)\(([^)]*)\)", block, re.S)}
    assert set(cf) == set(rf)
    for n in cf:
        assert cf[n] == rf[n], (n, cf[n], rf[n])
