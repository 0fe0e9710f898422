"""Crowd size sweep: frame time of the one-launch crowd kernel vs instances (30 000 verts / 200 bones each). If time = fixed + bytes / rate,
the fixed part is the launch ramp + palette staging + drain of ONE kernel and the rate is the write ceiling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
V, B = 30000, 200
mesh = synth.make_mesh(V, B)
base = [synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(64)]
c = rz.DeformContext(0)
c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"])
rows = []
for I in (64, 128, 256, 512, 1024, 2048):
    c.set_instances(I); c.set_pose(np.stack([base[i % 64] for i in range(I)]))
    t0 = time.time()
    while time.time() - t0 < 1.0: c.deform_n(100); c.sync()
    t = min((c.time_frames(100) for _ in range(3)), key=lambda t: t["frame_ms"])
    us, by = t["frame_ms"] * 1e3, t["algorithmic_bytes_per_frame"]
    rows.append((I, us, by))
    print("I=%5d: frame %.2f us  compulsory %.1f MB  -> %.0f GB/s = %.1f %% of 8 TB/s  (%s, grid %d x %d groups)" % (
        I, us, by / 1e6, by / us / 1e3, by / us / 1e3 / 80, c.kernel_name(), c.get_tuning("effective_grid"), (I + 7) // 8), flush=True)
(i0, t0, b0), (i1, t1, b1) = rows[2], rows[-1]
rate = (b1 - b0) / (t1 - t0) / 1e3
print("marginal rate between I=%d and I=%d: %.0f GB/s; fixed part at I=256: %.1f us" % (i0, i1, rate, t0 - b0 / rate / 1e3))
