"""A/B of two library builds on the default C4 frame (one box, alternating processes are run by the caller)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
rz.capi.LIB_PATH = os.environ["REZE_LIB"]
ctx = rz.DeformContext(0)
mesh = synth.make_mesh(30000, 200)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.set_instances(256)
worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], 200, seed=1000 + i) for i in range(256)])
ctx.set_pose(worlds)
for _ in range(30):
    ctx.deform_n(200); ctx.sync()
ts = sorted(ctx.time_frames(400)["frame_ms"] for _ in range(9))
print("%s frame median %.3f us (min %.3f max %.3f)" % (os.path.basename(rz.capi.LIB_PATH), ts[4] * 1e3, ts[0] * 1e3, ts[-1] * 1e3), flush=True)
