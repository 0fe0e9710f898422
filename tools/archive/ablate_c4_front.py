"""What does the FRONT of a crowd workgroup cost (staging the matrices + forming the palette rows, during which its CU stores
nothing)? C4 frame with and without it (dbg 8: tools-only ablation build, output is garbage), whole-palette and bone-subset forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
rz.capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libreze_deform_ablate.so")
ctx = rz.DeformContext(0)
mesh = synth.make_mesh(30000, 200)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.set_instances(256)
worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], 200, seed=1000 + i) for i in range(256)])
ctx.set_pose(worlds)
for _ in range(20):
    ctx.deform_n(200); ctx.sync()
for rep in range(2):
    for sub in (0, 1):
        row = []
        for dbg in (0, 8):
            ctx.set_tuning(inst_subsets=sub, dbg=dbg)
            row.append(sorted(ctx.time_frames(300)["frame_ms"] for _ in range(5))[2] * 1e3)
        print("%s: frame %.2f us, without the front %.2f us -> front = %.2f us" % ("bone subsets " if sub else "whole palette", row[0], row[1], row[0] - row[1]), flush=True)
ctx.set_tuning(dbg=0)
