"""Do back-to-back frames of a small shard gain from overlapping their launch ramps / tails? Two contexts holding the SAME 1/8 shard of
C5 (own streams, own output buffers) fed alternately, against one context running the same number of frames on one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
V, B, M = int(sys.argv[1]) if len(sys.argv) > 1 else 125184, 256, 64
mesh = synth.make_mesh(V, B)
deltas, w = synth.make_morphs_dense(V, M)
world = synth.make_pose(mesh["parents"], mesh["bind"], B, seed=3)
def make():
    c = rz.DeformContext(0)
    c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"]); c.upload_morphs_dense(deltas)
    c.set_pose(world, w); c.deform(); c.autotune(0)
    return c
a, b = make(), make()
t0 = time.time()
while time.time() - t0 < 2.5: a.deform_n(500); a.sync()
def one(n=2000):
    a.sync(); t = time.perf_counter(); a.deform_n(n); a.sync(); return (time.perf_counter() - t) / n * 1e6
def two(n=2000, chunk=1):
    a.sync(); b.sync(); t = time.perf_counter()
    for _ in range(n // (2 * chunk)): a.deform_n(chunk); b.deform_n(chunk)
    a.sync(); b.sync(); return (time.perf_counter() - t) / n * 1e6
print("V=%d: one stream %.2f us per frame (kernel %.2f us)" % (V, min(one() for _ in range(4)), a.time_frames(300)["deform_kernel_ms"] * 1e3))
for chunk in (1, 4, 25):
    two(200, chunk)
    print("   two streams, alternating every %d frame(s): %.2f us per frame" % (chunk, min(two(2000, chunk) for _ in range(4))))
