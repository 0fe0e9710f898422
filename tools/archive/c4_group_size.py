"""C4: what a smaller pose group costs at 512 threads per workgroup (one-launch form): the per-vertex loads + decode are amortised over G poses."""
import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
ctx = rz.DeformContext(0)
mesh = synth.make_mesh(30000, 200)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.set_instances(256)
worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], 200, seed=1000 + i) for i in range(256)])
ctx.set_pose(worlds)
for blk, G, cap in ((512, 8, 256), (512, 4, 256), (512, 4, 512), (512, 2, 256), (512, 6, 256), (512, 8, 256)):
    ctx.set_tuning(inst_block=blk, inst_loop=G, grid_cap=cap)
    t = min((ctx.time_frames(200) for _ in range(3)), key=lambda t: t["frame_ms"])
    print("block=%d G=%d(eff %d) cap=%d grid=%d: kernel %.2f us frame %.2f us" % (blk, G, ctx.get_tuning("effective_inst_group"), cap, ctx.get_tuning("effective_grid"), t["deform_kernel_ms"] * 1e3, t["frame_ms"] * 1e3), flush=True)
