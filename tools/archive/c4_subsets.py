"""C4 (256 x 30 000 / 200 bones / no morphs): bone-subset form of rz_skin_instances_kernel against the whole-palette form,
over poses per workgroup x workgroup size x total workgroups. Frame = everything a frame launches (HIP events, rz_time_frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
if os.environ.get("REZE_LIB"):             # an experimental build of the library (tools/_tmp/...)
    rz.capi.LIB_PATH = os.environ["REZE_LIB"]
    print("library:", rz.capi.LIB_PATH)
ctx = rz.DeformContext(0)
mesh = synth.make_mesh(30000, 200)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.set_instances(256)
worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], 200, seed=1000 + i) for i in range(256)])
ctx.set_pose(worlds)
for _ in range(20):
    ctx.deform_n(200); ctx.sync()          # clocks
quick = len(sys.argv) > 1 and sys.argv[1] in ("quick", "order")
rows = []


def run(sub, blk, il, cap, fast=-1, order=1):
    ctx.set_tuning(inst_subsets=sub, inst_block=blk, inst_loop=il, grid_cap=cap, fast=fast, inst_order=order)
    ts = [ctx.time_frames(300) for _ in range(3)]
    t = sorted(ts, key=lambda t: t["frame_ms"])[1]        # median of three
    rows.append((t["frame_ms"], sub, blk, il, cap, fast, t["deform_kernel_ms"]))
    print("order=%d subsets=%d block=%4d G=%2d(eff %2d) cap=%4d grid=%3d fast=%2d bones=%3d lds=%6d : kernel %.2f us frame %.2f us (%.1f %% of 8 TB/s at frame level)" % (
        order, ctx.get_tuning("effective_subsets"), blk, il, ctx.get_tuning("effective_inst_group"), cap, ctx.get_tuning("effective_grid"), fast,
        ctx.get_tuning("effective_subset_bones"), ctx.get_tuning("effective_inst_lds"),
        t["deform_kernel_ms"] * 1e3, t["frame_ms"] * 1e3, 188.69e6 / (t["frame_ms"] * 1e-3) / 8e12 * 100), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "mini":
    for cap in (256, 1024):
        run(1, 512, 8, cap)
    run(1, 512, 4, 1024); run(0, 512, 8, 256)
    sys.exit(0)
run(0, 512, 8, 256)                         # round 2's default
for blk in (512, 256) if not quick else (512,):
    for il in (8, 16, 32, 4) if not quick else (8, 4):
        for cap in (256, 512, 768, 1024, 1536, 2048) if not (len(sys.argv) > 1 and sys.argv[1] == "order") else ():
            run(1, blk, il, cap)
if len(sys.argv) > 1 and sys.argv[1] == "order":
    rows.clear()
    for order in (1, 0, 1):
        for il in (8, 4):
            for cap in (256, 512, 1024, 2048):
                if order == 0 and il == 4:
                    continue
                run(1, 512, il, cap, order=order)
if not quick:
    run(1, 1024, 16, 256); run(1, 1024, 32, 256)
    run(1, 512, 8, 256, fast=0); run(1, 256, 8, 512, fast=0)      # rz_prep_kernel in front: finished rows staged
best = min(rows)
print("best frame: %.2f us  subsets=%d block=%d G=%d cap=%d fast=%d (kernel %.2f us) -> %.1f %% of 8 TB/s" % (
    best[0] * 1e3, best[1], best[2], best[3], best[4], best[5], best[6] * 1e3, 188.69e6 / (best[0] * 1e-3) / 8e12 * 100))
