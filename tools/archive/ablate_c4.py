"""Ablation of the instanced (C4) skin kernel on one MI355X (tools-only build with the dbg switches, make ablate):
dbg 0 = full kernel, 1 = gathers + math without the output stream, 2 = output stream without gathers / math,
6 = full kernel without the palette staging at its start (garbage palettes), 7 = 2 + 6 (nothing but the mesh loads and the stores)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
# the ablation switches only exist in the tools-only build (make -C reze-engine_amd/csrc ablate)
rz.capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libreze_deform_ablate.so")
from reze_engine_amd import synth
ctx = rz.DeformContext(0)
mesh = synth.make_mesh(30000, 200)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.set_instances(256)
worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], 200, seed=1000 + i) for i in range(256)])
ctx.set_pose(worlds)
forms = [(256, 8, 512, 0), (1024, 8, 256, 0)] if len(sys.argv) < 2 else [(512, 8, 256, 0), (512, 8, 256, -1)]
if len(sys.argv) > 2:
    ctx.set_tuning(inst_order=int(sys.argv[2]))     # any argument: the default shape, prep-kernel and one-launch form
for blk, il, cap, fast in forms:
    for dbg in (0, 1, 2, 6, 7) + ((8,) if fast else ()):       # 8 (one-launch form): neither staging nor the in-kernel palette product
        ctx.set_tuning(inst_block=blk, inst_loop=il, grid_cap=cap, dbg=dbg, fast=fast)
        t = min((ctx.time_frames(200) for _ in range(3)), key=lambda t: t["deform_kernel_ms"])
        print("%s block=%d G=%d cap=%d dbg=%d kernel %.2f us frame %.2f us" % ("one-launch" if fast else "prep-form ", blk, il, cap, dbg, t["deform_kernel_ms"] * 1e3, t["frame_ms"] * 1e3), flush=True)
