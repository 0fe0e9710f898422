#!/bin/bash
# Disassemble every kernel of the given .hip files (product flags + extra flags) into one normalised text file per kernel:
#   tools/archive/isa_dump.sh <outdir> [-DFLAG ...] -- file.hip [file2.hip ...]
# (addresses and encodings stripped, so two builds of the same code compare equal with diff; used to check that a refactor leaves
# the instruction streams alone: tools/archive/isa_diff.py)
OUT=$1; shift
FL=()
while [ "$1" != "--" ]; do FL+=("$1"); shift; done; shift
D=$(cd "$(dirname "$0")/../reze-engine_amd/csrc" && pwd)
mkdir -p $OUT; T=$(mktemp -d)
for F in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -I$D "${FL[@]}" --offload-device-only -c "$F" -o $T/k.co || exit 1
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.elf || exit 1
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $T/k.elf | python3 -c '
import re, sys, subprocess, os
out = sys.argv[1]
cur, name = None, None
syms = {}
for line in sys.stdin:
    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
    if m:
        name = m.group(1); cur = syms.setdefault(name, []); continue
    if cur is None: continue
    t = re.sub(r"\s*//.*$", "", line.rstrip())
    t = re.sub(r"^\s+", "", t)
    if t: cur.append(t)
names = list(syms)
dem = subprocess.check_output(["c++filt"] + names).decode().splitlines()
for n, d in zip(names, dem):
    if not syms[n] or n.endswith(".kd"): continue
    d = d.replace("(anonymous namespace)::", "").replace("void ", "")
    d = re.sub(r"\(.*", "", d)
    fn = re.sub(r"[^A-Za-z0-9_<>,]", "", d.replace(" ", ""))
    open(os.path.join(out, fn + ".s"), "w").write("\n".join(syms[n]) + "\n")
' $OUT
done
rm -rf $T; ls $OUT | wc -l
