"""demo-shaped / 2 %-spread sparse frames: kernel time by how many CSR entry loads a lane keeps in flight (build-time RZ_SPARSE_INFLIGHT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
if os.environ.get("REZE_LIB"):
    rz.capi.LIB_PATH = os.environ["REZE_LIB"]
V, B, M = 28842, 349, 60
mesh = synth.make_mesh(V, B)
ctx = rz.DeformContext(0)
for name, gen in (("demo", lambda: synth.make_morphs_demo_shape(V, M)), ("sparse2", lambda: synth.make_morphs_sparse(V, M, density=0.02))):
    off, idx, d3, mw = gen()
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
    ctx.upload_morphs_sparse(off, idx, d3)
    ctx.set_pose(mesh["world"], mw)
    for _ in range(10):
        ctx.deform_n(500); ctx.sync()
    ts = sorted(ctx.time_frames(500)["deform_kernel_ms"] for _ in range(5))
    print("%s %-8s kernel %.2f us (min %.2f)" % (os.path.basename(rz.capi.LIB_PATH), name, ts[2] * 1e3, ts[0] * 1e3), flush=True)
