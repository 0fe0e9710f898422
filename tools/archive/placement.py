"""Where do a context's big buffers land, and does it matter? (NOTEBOOK.md R5.3) Fresh contexts of the C5 workload, one after the other in
ONE process: device addresses (4 KB pages) of the morph planes / outputs / rest geometry and the event-timed kernel, then the same with
the morph planes shifted inside a larger allocation (RZ_DENSE_OFFSET, tools-only build).   python tools/archive/placement.py [c5|shard]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
V, B, M = (1000000, 256, 64) if which == "c5" else (125184, 256, 64)
libs = {"product": rz.capi.load(), "variants": rz.capi.load(rz.capi.VARIANTS_LIB_PATH)}
if os.path.exists("tools/_tmp/old/libreze_deform_old.so"):
    libs["old"] = rz.capi.load("tools/_tmp/old/libreze_deform_old.so")
mesh = synth.make_mesh_range(max(V, 30000), B, 0, V)
d, mw = synth.make_morphs_dense_range(max(V, 30000), M, 0, V)
n = 200 if V >= 500000 else 1000
def run(lib, tag):
    c = rz.DeformContext(0, lib=libs[lib])
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"]); c.upload_morphs_dense(d)
    c.set_pose(mesh["world"], mw)
    for _ in range(4):
        c.deform_n(n // 2); c.sync()
    ts = sorted(c.time_frames(n)["deform_kernel_ms"] * 1e3 for _ in range(3))
    try:
        a = {k: c.get_tuning("addr_" + k) for k in ("dense", "out", "nrm", "geom")}
        where = " ".join("%s %#x (2MB frame offset %4d pages)" % (k, v, v % 512) for k, v in a.items())
    except Exception:
        where = "(library without address diagnostics)"
    print("%-10s %-22s kernel %.2f us | %s" % (lib, tag, ts[1], where), flush=True)
    c.close()
print("== fresh contexts, one after the other (GPU_MAX_HW_QUEUES=%s)" % os.environ.get("GPU_MAX_HW_QUEUES", "default"))
import torch
for i in range(6):
    for lib in libs:
        run(lib, "round %d" % i)
    if "extrastream" in sys.argv and i % 2 == 1:
        _keep = globals().setdefault("_keep", []); _keep.append(torch.cuda.Stream())      # shifts which hardware queue the next context's streams map to
        print("(one extra stream created)")
if "nooffset" in sys.argv:
    sys.exit(0)
print("== morph planes shifted inside a larger allocation (variants build)")
for off in (0, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20):
    os.environ["RZ_DENSE_OFFSET"] = str(off)
    for rep in range(2):
        run("variants", "dense offset %d" % off)
os.environ.pop("RZ_DENSE_OFFSET", None)
