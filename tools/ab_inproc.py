"""Same-process A/B of several builds of the library (they are linked -Bsymbolic-functions and loaded RTLD_LOCAL, so they live side by
side): every build gets its own context on the same data, then the builds are timed ALTERNATELY, round after round — the only kind
of comparison that separates code from box (profiles/r4_box_variance.txt).
  python tools/ab_inproc.py <workload>[,<workload>...] <rounds> name=path [name=path ...]
workloads: c5 shard c3 c2 demo c4      kernel / frame us by events (rz_time_frames), median over the rounds + every round's value.
<rounds> = N: one context per build, timed N times alternately; fN: a FRESH context (new allocations, new upload) per build and round —
where a buffer lands in HBM moves a kernel by up to ~0.7 % (two contexts of the SAME build differ by that much), so the question
"is build B slower than build A" wants fresh placements on both sides."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth

workloads = sys.argv[1].split(",")
fresh = sys.argv[2].startswith("f")
rounds = int(sys.argv[2].lstrip("f"))
builds = [a.split("=", 1) for a in sys.argv[3:]]
libs = [(n, rz.capi.load(p)) for n, p in builds]
SHAPES = {"c5": (1000000, 256, 64, 1), "shard": (125184, 256, 64, 1), "c3": (30000, 200, 64, 1), "c2": (30000, 200, 0, 1), "c4": (30000, 200, 0, 256)}
for wl in workloads:
    ctxs = []
    if wl in ("demo", "sparse2"):
        V, B, M, I = 28842, 349, 60, 1
        mesh = synth.make_mesh(V, B)
        off, idx, d3, mw = synth.make_morphs_demo_shape(V, M) if wl == "demo" else synth.make_morphs_sparse(V, M, density=0.02)
        d = None
    else:
        V, B, M, I = SHAPES[wl]
        mesh = synth.make_mesh_range(max(V, 30000), B, 0, V)
        d, mw = synth.make_morphs_dense_range(max(V, 30000), M, 0, V) if M else (None, None)
    worlds = mesh["world"] if I == 1 else np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)])
    def make(lib):
        c = rz.DeformContext(0, lib=lib)
        c.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); c.upload_skeleton(mesh["inv_bind"])
        if wl in ("demo", "sparse2"):
            c.upload_morphs_sparse(off, idx, d3)
        else:
            c.upload_morphs_dense(d)
        if I > 1:
            c.set_instances(I)
        c.set_pose(worlds, mw)
        return c
    n = 200 if V >= 500000 else (500 if I > 1 else 1000)
    res = {name: [] for name, _ in libs}
    if fresh:
        for r in range(rounds):
            order = libs if r % 2 == 0 else libs[::-1]
            for name, lib in order:
                c = make(lib)
                for _ in range(4):
                    c.deform_n(n // 2); c.sync()
                t = c.time_frames(n)
                res[name].append((t["deform_kernel_ms"] * 1e3, t["frame_ms"] * 1e3))
                c.close()
        ctxs = [(name, None) for name, _ in libs]
    else:
        for name, lib in libs:
            ctxs.append((name, make(lib)))
        for name, c in ctxs:
            for _ in range(6):
                c.deform_n(n // 2); c.sync()
        for r in range(rounds):
            order = ctxs if r % 2 == 0 else ctxs[::-1]          # alternate the order too
            for name, c in order:
                c.deform_n(n // 4); c.sync()
                t = c.time_frames(n)
                res[name].append((t["deform_kernel_ms"] * 1e3, t["frame_ms"] * 1e3))
    base = np.median([k for k, _ in res[ctxs[0][0]]])
    for name, _ in ctxs:
        ks = [k for k, _ in res[name]]; fs = [f for _, f in res[name]]
        print("%-8s %-10s kernel median %.3f us (%+.2f %% vs %s) frame median %.3f | kernel rounds: %s" % (
            wl, name, np.median(ks), (np.median(ks) / base - 1) * 100, ctxs[0][0], np.median(fs), " ".join("%.2f" % k for k in ks)), flush=True)
    # paired: in how many rounds was build k faster than the first build?
    for name, _ in ctxs[1:]:
        wins = sum(1 for a, b in zip(res[name], res[ctxs[0][0]]) if a[0] < b[0])
        print("%-8s %-10s faster than %s in %d of %d rounds" % (wl, name, ctxs[0][0], wins, rounds), flush=True)
    for _, c in ctxs:
        if c is not None:
            c.close()
    del d
