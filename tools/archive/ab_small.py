"""A/B of two library builds on the small single-character frames (C2, demo-shaped sparse, 2 %-spread sparse): kernel time by events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
rz.capi.LIB_PATH = os.environ["REZE_LIB"]
ctx = rz.DeformContext(0)
out = []
for name, V, B, gen in (("c2", 30000, 200, None), ("demo", 28842, 349, lambda V: synth.make_morphs_demo_shape(V, 60)), ("sparse2", 28842, 349, lambda V: synth.make_morphs_sparse(V, 60, density=0.02))):
    mesh = synth.make_mesh(V, B)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
    mw = None
    if gen:
        off, idx, d3, mw = gen(V)
        ctx.upload_morphs_sparse(off, idx, d3)
    ctx.set_pose(mesh["world"], mw)
    for _ in range(10):
        ctx.deform_n(500); ctx.sync()
    ts = sorted(ctx.time_frames(1000)["deform_kernel_ms"] for _ in range(7))
    out.append("%s %.3f" % (name, ts[3] * 1e3))
print(os.path.basename(rz.capi.LIB_PATH), " | ".join(out), flush=True)
