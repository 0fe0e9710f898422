"""Per-frame-pose loop of a 1/8 shard of C5 (rz_set_pose + rz_deform per frame) for one library build (REZE_LIB)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
if os.environ.get("REZE_LIB"):
    rz.capi.LIB_PATH = os.environ["REZE_LIB"]
ctx = rz.DeformContext(0)
V = 125184
mesh = synth.make_mesh_range(1000000, 256, 0, V); deltas, mw = synth.make_morphs_dense_range(1000000, 64, 0, V)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.upload_morphs_dense(deltas); ctx.set_pose(mesh["world"], mw)
t0 = time.time()
while time.time() - t0 < 2.0: ctx.deform_n(500); ctx.sync()
L, h = rz.capi.load(), ctx._h
fp = ctypes.POINTER(ctypes.c_float)
w32, m32 = np.ascontiguousarray(mesh["world"], np.float32), np.ascontiguousarray(mw, np.float32)
wp, mp = w32.ctypes.data_as(fp), m32.ctypes.data_as(fp)
n, res = 3000, []
for rep in range(5):
    ctx.sync(); t0 = time.perf_counter(); ctx.deform_n(n); ctx.sync(); replay = (time.perf_counter() - t0) / n
    for _ in range(300): L.rz_set_pose(h, wp, mp); L.rz_deform(h)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): L.rz_set_pose(h, wp, mp); L.rz_deform(h)
    ctx.sync(); live = (time.perf_counter() - t0) / n
    res.append((live - replay, replay, live))
res.sort()
print("%s: replay %.2f us, set_pose + deform %.2f us: +%.2f us (median of 5; min +%.2f)" % (os.path.basename(rz.capi.LIB_PATH), res[2][1] * 1e6, res[2][2] * 1e6, res[2][0] * 1e6, res[0][0] * 1e6), flush=True)
