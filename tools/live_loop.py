"""Per-frame (PCIe-inclusive) loop of a single character: rz_set_pose + rz_deform per frame vs the resident-pose replay,
for C5 and one 1/8 shard of it. The difference is what a frame pays for its 16 KB of world matrices + 256 B of weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
ctx = rz.DeformContext(0)
for V in (1000000, 125184):
    mesh = synth.make_mesh_range(1000000, 256, 0, V); deltas, mw = synth.make_morphs_dense_range(1000000, 64, 0, V)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
    ctx.upload_morphs_dense(deltas); ctx.set_pose(mesh["world"], mw)
    ctx.autotune()
    t0 = time.time()
    while time.time() - t0 < 2.5: ctx.deform_n(500); ctx.sync()
    n = 2000
    res = []
    import ctypes
    L, h = rz.capi.load(), ctx._h
    fp = ctypes.POINTER(ctypes.c_float)
    w32, m32 = np.ascontiguousarray(mesh["world"], np.float32), np.ascontiguousarray(mw, np.float32)
    wp, mp = w32.ctypes.data_as(fp), m32.ctypes.data_as(fp)
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter(); ctx.deform_n(n); ctx.sync(); replay = (time.perf_counter() - t0) / n
        ctx.set_tuning(pose_prefetch=0)
        for _ in range(200): L.rz_set_pose(h, wp, mp); L.rz_deform(h)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(n): L.rz_set_pose(h, wp, mp); L.rz_deform(h)
        ctx.sync(); live_nopf = (time.perf_counter() - t0) / n
        ctx.set_tuning(pose_prefetch=-1)
        for _ in range(200): L.rz_set_pose(h, wp, mp); L.rz_deform(h)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(n): L.rz_set_pose(h, wp, mp); L.rz_deform(h)        # raw C ABI calls: no numpy conversions in the loop
        ctx.sync(); live = (time.perf_counter() - t0) / n
        staged = ctx.get_tuning("pose_staged")
        t0 = time.perf_counter()
        for _ in range(n): L.rz_set_pose(h, wp, mp)
        ctx.sync(); up_only = (time.perf_counter() - t0) / n
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(n): L.rz_deform(h)
        ctx.sync(); single = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n): L.rz_deform(h)
        host_only = (time.perf_counter() - t0) / n; ctx.sync()
        res.append((replay, live, up_only, single, host_only, live_nopf, staged))
    r = min(res)
    print("V=%7d: replay (deform_n) %.2f us/frame | rz_deform per frame %.2f us (host side of the call %.2f us) | set_pose + deform %.2f us/frame (+%.2f over replay, +%.2f over per-frame deform; last pose staged by the previous frame: %d) | the same with pose_prefetch = 0: %.2f us/frame | set_pose alone %.2f us"
          % (V, r[0] * 1e6, r[3] * 1e6, r[4] * 1e6, r[1] * 1e6, (r[1] - r[0]) * 1e6, (r[1] - r[3]) * 1e6, r[6], r[5] * 1e6, r[2] * 1e6), flush=True)
