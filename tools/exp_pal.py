import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
ctx = rz.DeformContext(0)
for V in (125952, 1000000):
    mesh = synth.make_mesh(V, 256); deltas, mw = synth.make_morphs_dense(V, 64)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
    ctx.upload_morphs_dense(deltas); ctx.set_pose(mesh["world"], mw)
    for dbg in (0, 3, 0, 3):
        ctx.set_tuning(dbg=dbg)
        best = min(ctx.time_frames(300 if V < 500000 else 60)["frame_ms"] for _ in range(4))
        print("V=%d dbg=%d frame %.4f ms" % (V, dbg, best))
