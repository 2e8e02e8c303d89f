#!/bin/bash
# Which kernels' machine code differs between a committed revision and the working tree?  (CPU only: hipcc cross-compiles.)
#   bash profiles/micro/isa_diff.sh [REV]      (default: HEAD)
# Compiles josefine_gpu.hip device-only for gfx950 from both, disassembles, and compares every kernel's instruction
# stream (addresses and symbol offsets stripped).  Used to show that an opt-in path leaves the default kernels as they were.
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=$(mktemp -d /tmp/jg_isa_XXXX)
mkdir -p $TMP/base
git -C $ROOT archive $REV josefine_amd/csrc include | tar -x -C $TMP/base
OBJDUMP=/opt/rocm/lib/llvm/bin/llvm-objdump
for side in base work; do
  if [ $side = base ]; then SRC=$TMP/base/josefine_amd/csrc; else SRC=$ROOT/josefine_amd/csrc; fi
  (cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c josefine_gpu.hip -o $TMP/$side.o 2>/dev/null)
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hip-amdgcn-amd-amdhsa--gfx950 --input=$TMP/$side.o --output=$TMP/$side.elf
  $OBJDUMP -d --no-show-raw-insn $TMP/$side.elf > $TMP/$side.s
done
python3 - $TMP/base.s $TMP/work.s <<'PY'
import re, sys, subprocess, hashlib
def kernels(path):
    out, cur = {}, None
    for ln in open(path):
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', ln)
        if m:
            cur = m.group(1); out[cur] = []; continue
        if cur and ln.strip():
            ins = re.sub(r'^\s*[0-9a-f]+:\s*', '', ln.rstrip())
            ins = re.sub(r'//.*$', '', ins).strip()
            ins = re.sub(r'<[^>]+>', '<>', ins)          # branch targets by symbol+offset
            out[cur].append(ins)
    return out
a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
def dem(n):
    return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '').split('(')[0]
same = [k for k in a if k in b and a[k] == b[k]]
diff = [k for k in a if k in b and a[k] != b[k]]
print(f"{len(same)} kernels / functions identical, {len(diff)} differ, {len([k for k in b if k not in a])} new, {len([k for k in a if k not in b])} gone")
for k in diff: print("DIFFERS:", dem(k), len(a[k]), "->", len(b[k]), "instructions")
for k in b:
    if k not in a: print("new:", dem(k))
for k in a:
    if k not in b: print("gone:", dem(k))
PY
rm -rf $TMP
