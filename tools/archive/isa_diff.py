"""Compare two directories of normalised per-kernel disassembly (tools/archive/isa_dump.sh): for every kernel of the NEW directory, the kernel it
replaces in the OLD one (rz_deform_kernel<S,U,MODE,NT,NTS,GEO,FAST> was split into rz_deform_dense_kernel<S,U,NT,NTS,GEO,FAST> and
rz_deform_small_kernel<S,MODE,NTS,GEO,FAST>), identical or not, instruction counts, and the mnemonic histogram differences."""
import collections, difflib, os, re, sys
old, new = sys.argv[1], sys.argv[2]
def old_name(n):
    m = re.match(r"rz_deform_dense_kernel<(\d+),(\d+),(\w+),(\w+),(\w+),(\w+)>", n)
    if m: return "rz_deform_kernel<%s,%s,1,%s,%s,%s,%s>" % m.groups()
    m = re.match(r"rz_deform_small_kernel<(\d+),(\d+),(\w+),(\w+),(\w+)>", n)
    if m: return "rz_deform_kernel<%s,1,%s,false,%s,%s,%s>" % m.groups()
    return n
same = diff = 0
for f in sorted(os.listdir(new)):
    n = f[:-2]
    o = os.path.join(old, old_name(n) + ".s")
    if not os.path.exists(o):
        print("%-70s NEW (no counterpart)" % n); continue
    a, b = open(o).read().splitlines(), open(os.path.join(new, f)).read().splitlines()
    if a == b:
        same += 1; print("%-70s identical (%d instructions)" % (n, len(a))); continue
    diff += 1
    ha = collections.Counter(l.split()[0] for l in a if l and not l.endswith(":")); hb = collections.Counter(l.split()[0] for l in b if l and not l.endswith(":"))
    d = {k: hb[k] - ha[k] for k in set(ha) | set(hb) if hb[k] != ha[k]}
    sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
    print("%-70s DIFFERS: %d -> %d instructions, similarity %.4f, histogram delta %s" % (n, len(a), len(b), sm.ratio(), dict(sorted(d.items())) if len(d) < 16 else "%d mnemonics" % len(d)))
print("%d identical, %d differ" % (same, diff))
