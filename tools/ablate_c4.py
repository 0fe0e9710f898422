import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
ctx = rz.DeformContext(0)
mesh = synth.make_mesh(30000, 200)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.set_instances(256)
worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], 200, seed=1000 + i) for i in range(256)])
ctx.set_pose(worlds)
for il, cap, dbg in ((8, 512, 0), (8, 512, 1), (8, 512, 2), (4, 2048, 0), (4, 2048, 1), (4, 2048, 2)):
    ctx.set_tuning(inst_loop=il, grid_cap=cap, dbg=dbg)
    for rep in range(2):
        t = ctx.time_frames(100)
        ctx.sync(); t0 = time.perf_counter(); ctx.deform_n(200); ctx.sync(); wall = (time.perf_counter() - t0) / 200 * 1e3
        print(il, cap, "dbg", dbg, rep, "frame %.4f kernel %.4f prep %.4f wall %.4f" % (t["frame_ms"], t["deform_kernel_ms"], t["prep_kernel_ms"], wall))
