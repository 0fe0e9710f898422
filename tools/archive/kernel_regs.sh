#!/bin/bash
# Register / scratch use of every kernel of a .hip file, read from the code object's metadata (no GPU needed):
#   tools/archive/kernel_regs.sh [file.hip] [extra flags...]   -> "<vgpr> <sgpr> <sgpr spills> <scratch B> <kernel>" per kernel, sorted by VGPRs
D=$(cd "$(dirname "$0")/../../reze-engine_amd/csrc" && pwd)
F=${1:-$D/kernels/deform_dense.hip}; shift
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -I$D "$@" --offload-device-only -c "$F" -o $T/k.co || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.elf || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.elf | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.match(r"\s+(- )?\.(\w+):\s+(.*)", line)
    if not m: continue
    if m.group(1) and m.group(2) == "agpr_count":
        cur = {}; rows.append(cur)
    if cur is not None: cur[m.group(2)] = m.group(3).strip()
rows = [r for r in rows if "name" in r and "vgpr_count" in r]
names = subprocess.check_output(["c++filt"] + [r["name"] for r in rows]).decode().splitlines()
for r, n in sorted(zip(rows, names), key=lambda t: -int(t[0]["vgpr_count"])):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    print("%4s vgpr %4s sgpr %3s sgpr-spills %4s B scratch  %s" % (r["vgpr_count"], r["sgpr_count"], r.get("sgpr_spill_count"), r.get("private_segment_fixed_size"), n))
'
rm -rf $T
