"""Generates a humanoid-shaped synthetic PMX (30 k vertices, 200 bones, 14 levels, 30 sparse vertex morphs) and a VMD
(100 keyed bones with translations, 10 keyed morphs) under /tmp and runs tools/node_frame_bench.js on them."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pmx_synth import write_pmx, write_vmd
import numpy as np
os.makedirs("/tmp/nb", exist_ok=True)
open("/tmp/nb/m.pmx", "wb").write(write_pmx(V=30000, B=200, n_vertex_morphs=30, max_depth=14))
rng = np.random.default_rng(1)
keys = []
for b in range(0, 200, 2):
    for f in range(0, 61, 10):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        keys.append(("bone%d" % b, f, tuple(q), tuple(rng.normal(size=3) * 0.1)))
open("/tmp/nb/a.vmd", "wb").write(write_vmd(keys, [("v%d" % i, f, float(rng.random())) for i in range(10) for f in (0, 30, 60)]))
print(subprocess.check_output(["node", os.path.join(ROOT, "tools", "node_frame_bench.js"), "/tmp/nb/m.pmx", "/tmp/nb/a.vmd", "3000"]).decode())
