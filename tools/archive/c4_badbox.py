"""Is this one of the boxes where more than one round of crowd workgroups is slow (NOTEBOOK.md R3.1)? If so, compare the workgroup
orders there. Prints BOX=good|bad first."""
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
for _ in range(20):
    ctx.deform_n(200); ctx.sync()


def t(cap, order, il=8, blk=512):
    ctx.set_tuning(inst_subsets=1, inst_block=blk, inst_loop=il, grid_cap=cap, inst_order=order)
    return sorted(ctx.time_frames(300)["frame_ms"] for _ in range(3))[1] * 1e3


a, b = t(256, 1), t(512, 1)
bad = b > 1.12 * a
print("BOX=%s  one round %.2f us, two rounds %.2f us" % ("bad" if bad else "good", a, b), flush=True)
if bad or (len(sys.argv) > 1 and sys.argv[1] == "force"):
    for rep in range(2):
        for order in (1, 0):
            print("order %d: " % order + "  ".join("cap %4d G %d: %.2f us" % (cap, il, t(cap, order, il)) for il in (8, 4) for cap in (512, 1024)), flush=True)
