"""Would two independent half-crowds on two streams beat one crowd kernel? Two contexts (own streams) with 128 instances each,
both queues filled, against one context with 256 instances: same total work per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
V, B, I = 30000, 200, 256
mesh = synth.make_mesh(V, B)
worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)])
def make(n, w, cap=None):
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"])
    c.set_instances(n); c.set_pose(w)
    if cap: c.set_tuning(grid_cap=cap)
    return c
one = make(256, worlds)
t0 = time.time()
while time.time() - t0 < 2.5: one.deform_n(500); one.sync()
def run_one(n=1000):
    one.sync(); t = time.perf_counter(); one.deform_n(n); one.sync(); return (time.perf_counter() - t) / n * 1e6
print("one context, 256 instances: %.2f us per frame" % min(run_one() for _ in range(4)))
for cap in (128, 256):
    a, b = make(128, worlds[:128], cap), make(128, worlds[128:], cap)
    def run_two(n=1000, chunk=50):
        a.sync(); b.sync(); t = time.perf_counter()
        for _ in range(n // chunk): a.deform_n(chunk); b.deform_n(chunk)
        a.sync(); b.sync(); return (time.perf_counter() - t) / n * 1e6
    run_two(200)
    print("two contexts x 128 instances (grid_cap %d each, %s): %.2f us per frame of 256" % (cap, a.kernel_name(), min(run_two() for _ in range(4))))
    ta = a.time_frames(300)["frame_ms"] * 1e3
    print("   one half alone: %.2f us" % ta)
    a.close(); b.close()
for parts in (4,):
    cs = [make(I // parts, worlds[k * (I // parts):(k + 1) * (I // parts)], 256 // parts) for k in range(parts)]
    def run_n(n=1000, chunk=50):
        for c in cs: c.sync()
        t = time.perf_counter()
        for _ in range(n // chunk):
            for c in cs: c.deform_n(chunk)
        for c in cs: c.sync()
        return (time.perf_counter() - t) / n * 1e6
    run_n(200)
    print("%d contexts x %d instances: %.2f us per frame of 256" % (parts, I // parts, min(run_n() for _ in range(4))))
