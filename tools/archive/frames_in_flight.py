import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
V, B, M = int(sys.argv[1]) if len(sys.argv) > 1 else 125184, 256, 64
mesh = synth.make_mesh(V, B); deltas, w = synth.make_morphs_dense(V, M)
world = synth.make_pose(mesh["parents"], mesh["bind"], B, seed=3)
a = rz.DeformContext(0)
a.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); a.upload_skeleton(mesh["inv_bind"]); a.upload_morphs_dense(deltas)
a.set_pose(world, w); a.deform(); a.autotune(0)
cs = [a] + [a.fork() for _ in range(3)]
for c in cs[1:]: c.set_pose(world, w)
t0 = time.time()
while time.time() - t0 < 2.5: a.deform_n(500); a.sync()
L = a._L
def run(k, n=3000):
    for c in cs: c.sync()
    t = time.perf_counter()
    for f in range(n): L.rz_deform(cs[f % k]._h)
    for c in cs: c.sync()
    return (time.perf_counter() - t) / n * 1e6
for k in (1, 2, 3, 4):
    run(k, 300)
    print("V=%d: %d frame(s) in flight: %.2f us per frame" % (V, k, min(run(k) for _ in range(4))))
