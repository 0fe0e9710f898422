"""C4 and device-animated C4 frames with and without the overlapped-front protocol (fronts of frame f+1 on the upload
stream under the skin kernel of frame f), HIP-event timing of back-to-back frames + wall-clock per-frame loops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
V, B, I = 30000, 200, 256
ctx = rz.DeformContext(0)
mesh = synth.make_mesh(V, B)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
ctx.set_instances(I)
ctx.upload_skeleton_topology(mesh["parents"], mesh["bind"])
worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)])
rng = np.random.default_rng(1)
quats = rng.normal(size=(I, B, 4)).astype(np.float32); quats /= np.linalg.norm(quats, axis=2, keepdims=True)
nk = 8
kq = rng.normal(size=(B, nk, 4)).astype(np.float32); kq /= np.linalg.norm(kq, axis=2, keepdims=True)
ctx.upload_animation(np.arange(B), np.arange(B + 1) * nk, np.tile(np.arange(nk) * 10.0, B), kq, (rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.2)
frames = rng.random(I).astype(np.float32) * 70
kinds = {"host world (prep + skin)": lambda: ctx.set_pose(worlds), "local rotations (fk + skin)": lambda: ctx.set_pose_local(quats),
         "sampled on device (sample + fk + skin)": lambda: ctx.set_pose_sampled(frames)}
for name, put in kinds.items():
    for ov in (0, 1):
        ctx.set_tuning(overlap=ov)
        put()
        t = min((ctx.time_frames(300) for _ in range(3)), key=lambda t: t["frame_ms"])
        # wall-clock loops: replay (deform_n) and per-frame upload + deform
        ctx.deform_n(50); ctx.sync()
        t0 = time.perf_counter(); ctx.deform_n(500); ctx.sync(); replay = (time.perf_counter() - t0) / 500
        for _ in range(300): put(); ctx.deform()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(300): put(); ctx.deform()
        ctx.sync(); live = (time.perf_counter() - t0) / 300
        print("%-40s overlap=%d: frame(events) %.2f us  kernel %.2f us  front %.2f us | replay wall %.2f us | upload+deform wall %.2f us  -> %.1f %% of 8 TB/s at frame level" % (
            name, ov, t["frame_ms"] * 1e3, t["deform_kernel_ms"] * 1e3, t["prep_kernel_ms"] * 1e3, replay * 1e6, live * 1e6, 188.69e6 / (t["frame_ms"] * 1e-3) / 8e12 * 100), flush=True)
