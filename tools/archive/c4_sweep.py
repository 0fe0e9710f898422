"""C4 (256 x 30 000 / 200 bones / no morphs) launch-shape sweep of rz_skin_instances_kernel on one MI355X:
workgroup size x poses per workgroup x total workgroups, kernel and whole-frame times from HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
ctx = rz.DeformContext(0)
mesh = synth.make_mesh(30000, 200)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.set_instances(256)
worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], 200, seed=1000 + i) for i in range(256)])
ctx.set_pose(worlds)
rows = []
for blk, loops, caps in ((256, (8, 4), (512, 1024)), (512, (8, 16), (256, 512)), (1024, (8, 16), (256, 512))):
    for il in loops:
        for cap in caps:
            for nts in (0, 1) if (blk, il, cap) in ((256, 8, 512), (1024, 8, 256)) else (0,):
                ctx.set_tuning(inst_block=blk, inst_loop=il, grid_cap=cap, nt_store=nts)
                t = min((ctx.time_frames(200) for _ in range(3)), key=lambda t: t["frame_ms"])
                rows.append((t["frame_ms"], blk, il, cap, nts, t["deform_kernel_ms"], t["prep_kernel_ms"]))
                print("block=%4d G=%2d(eff %2d) cap=%4d grid=%3d nts=%d : kernel %.2f us frame %.2f us prep %.2f us" % (
                    blk, il, ctx.get_tuning("effective_inst_group"), cap, ctx.get_tuning("effective_grid"), nts,
                    t["deform_kernel_ms"] * 1e3, t["frame_ms"] * 1e3, t["prep_kernel_ms"] * 1e3), flush=True)
# one-launch frames: palettes formed inside the skin kernel (fast = 1), no prep kernel, no launch boundary
for blk, il, cap in ((256, 8, 512), (256, 6, 512), (256, 4, 1024), (512, 8, 256), (512, 8, 512), (1024, 8, 256), (1024, 12, 256)):
    ctx.set_tuning(inst_block=blk, inst_loop=il, grid_cap=cap, nt_store=0, fast=1)
    t = min((ctx.time_frames(200) for _ in range(3)), key=lambda t: t["frame_ms"])
    rows.append((t["frame_ms"], blk, il, cap, 0, t["deform_kernel_ms"], t["prep_kernel_ms"]))
    print("ONE-LAUNCH block=%4d G=%2d(eff %2d) cap=%4d grid=%3d : kernel %.2f us frame %.2f us" % (
        blk, il, ctx.get_tuning("effective_inst_group"), cap, ctx.get_tuning("effective_grid"), t["deform_kernel_ms"] * 1e3, t["frame_ms"] * 1e3), flush=True)
ctx.set_tuning(fast=-1)
best = min(rows)
print("best frame: %.2f us at block=%d G=%d cap=%d nts=%d (kernel %.2f us); compulsory 188.69 MB -> %.1f %% of 8 TB/s (frame), %.1f %% (kernel)" % (
    best[0] * 1e3, best[1], best[2], best[3], best[4], best[5] * 1e3, 188.69e6 / (best[0] * 1e-3) / 8e12 * 100, 188.69e6 / (best[5] * 1e-3) / 8e12 * 100))
# a mesh whose influence types are clustered like a real PMX model (sorted by influence count): the wave-uniform skipping pays here
w = mesh["weights"].astype(np.int32)
order = np.argsort((w > 0).sum(axis=1), kind="stable")
ctx.upload_mesh(mesh["pos"][order], mesh["nrm"][order], mesh["joints"][order], mesh["weights"][order]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.set_instances(256); ctx.set_pose(worlds)
ctx.set_tuning(inst_block=best[1], inst_loop=best[2], grid_cap=best[3], nt_store=best[4])
t = min((ctx.time_frames(200) for _ in range(3)), key=lambda t: t["frame_ms"])
print("same shape, vertices clustered by influence count: kernel %.2f us frame %.2f us" % (t["deform_kernel_ms"] * 1e3, t["frame_ms"] * 1e3))
