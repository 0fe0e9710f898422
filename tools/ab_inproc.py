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
SHAPES = {"c5": (1000000, 256, 64, 1), "shard": (125184, 256, 64, 1), "shard2": (500224, 256, 64, 1), "shard4": (250112, 256, 64, 1), "c3": (30000, 200, 64, 1), "c2": (30000, 200, 0, 1), "c4": (30000, 200, 0, 256)}
for wl in workloads:
    ctxs = []
    anim = None
    if "-" in wl:                       # sampled-c2, sampled-demo, local-c2 ...: the pose comes from the device-side sampler / a local pose
        anim, wl = wl.split("-")
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
        if anim:
            c.upload_skeleton_topology(mesh["parents"], mesh["bind"])
            rng = np.random.default_rng(1)
            if anim == "sampled":
                nk = 8
                kq = rng.normal(size=(B, nk, 4)).astype(np.float32); kq /= np.linalg.norm(kq, axis=2, keepdims=True)
                extra = {}
                if M:
                    extra = dict(mkey_off=np.arange(M + 1) * 2, mkey_frame=np.tile(np.array([0.0, 70.0], np.float32), M), mkey_weight=np.repeat(mw, 2),
                                 feed_off=np.arange(M + 1), feed_track=np.arange(M), feed_ratio=np.ones(M, np.float32))
                c.upload_animation(np.arange(B), np.arange(B + 1) * nk, np.tile(np.arange(nk) * 10.0, B), kq, (rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.2,
                                   np.tile(np.array([20] * 8 + [107] * 8, np.uint8), B * nk), **extra)
                c.set_pose_sampled((13.5 + 0.37 * np.arange(I)).astype(np.float32) % 70.0)
            else:
                q = rng.normal(size=(I, B, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=2, keepdims=True)
                c.set_pose_local(q if I > 1 else q[0], mw)
        else:
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
        print("%-12s %-10s kernel median %.3f us (%+.2f %% vs %s) frame median %.3f | kernel rounds: %s" % (
            ((anim + "-") if anim else "") + wl, name, np.median(ks), (np.median(ks) / base - 1) * 100, ctxs[0][0], np.median(fs), " ".join("%.2f" % k for k in ks)), flush=True)
    # paired: in how many rounds was build k faster than the first build?
    for name, _ in ctxs[1:]:
        wins = sum(1 for a, b in zip(res[name], res[ctxs[0][0]]) if a[0] < b[0])
        print("%-12s %-10s faster than %s in %d of %d rounds" % (((anim + "-") if anim else "") + wl, name, ctxs[0][0], wins, rounds), flush=True)
    for _, c in ctxs:
        if c is not None:
            c.close()
    del d
