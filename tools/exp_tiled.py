import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
import oracle
ctx = rz.DeformContext(0)
for V in (1000000, 125952):
    mesh = synth.make_mesh(V, 256)
    deltas, mw = synth.make_morphs_dense(V, 64)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
    ref = None
    for tiled in (0, 1, 0, 1):
        ctx.set_tuning(morph_tiled=tiled)
        ctx.upload_morphs_dense(deltas); ctx.set_pose(mesh["world"], mw)
        ctx.deform(); p, n = ctx.read()
        if ref is None: ref = p
        same = bool(np.array_equal(p, ref))
        best = min(ctx.time_frames(60 if V > 500000 else 300)["frame_ms"] for _ in range(4))
        print("V=%d tiled=%d frame %.4f ms  bit-identical-to-planes=%s" % (V, tiled, best, same))
