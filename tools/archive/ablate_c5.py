"""Ablation of the fused morph+skin kernel (C5 and its 1/8 shard) on one MI355X. dbg 0 = full; 3 = no palette;
4 = morph phase only (no skin phase, no stores); 5 = skin phase without its output stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
# the ablation switches only exist in the tools-only build (make -C reze-engine_amd/csrc ablate)
rz.capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libreze_deform_ablate.so")
from reze_engine_amd import synth
ctx = rz.DeformContext(0)
for V in (1000000, 125184):
    mesh = synth.make_mesh(V, 256); deltas, mw = synth.make_morphs_dense(V, 64)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
    ctx.upload_morphs_dense(deltas); ctx.set_pose(mesh["world"], mw)
    for dbg, oc in ((0, 0), (0, -1), (0, 256), (5, 0), (0, 0), (0, -1)):
        ctx.set_tuning(dbg=dbg, out_cap=oc)
        best = min(ctx.time_frames(300 if V < 500000 else 60)["frame_ms"] for _ in range(4))
        print("V=%d dbg=%d out_cap=%d frame %.4f ms" % (V, dbg, oc, best))
