"""Round-4 A/B of one library build (REZE_LIB, default the product) on the frames rounds 4's kernel changes touch:
  small single-character frames by events (C2, demo-shaped sparse, 2 %-spread sparse: kernel time of a resident replay),
  device-animated single characters (sampled / local poses: per-frame loops through the C ABI, and the resident replay)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
if os.environ.get("REZE_LIB"):
    rz.capi.LIB_PATH = os.environ["REZE_LIB"]
tag = os.path.basename(rz.capi.LIB_PATH)
which = sys.argv[1:] or ["small", "anim"]
if "small" in which:
    ctx = rz.DeformContext(0)
    out = []
    for name, V, B, gen in (("c2", 30000, 200, None), ("demo", 28842, 349, lambda V: synth.make_morphs_demo_shape(V, 60)), ("sparse2", 28842, 349, lambda V: synth.make_morphs_sparse(V, 60, density=0.02))):
        mesh = synth.make_mesh(V, B)
        ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
        mw = None
        if gen:
            off, idx, d3, mw = gen(V)
            ctx.upload_morphs_sparse(off, idx, d3)
        else:
            ctx.upload_morphs_dense(None)
        ctx.set_pose(mesh["world"], mw)
        for _ in range(10):
            ctx.deform_n(500); ctx.sync()
        ts = sorted(ctx.time_frames(1000)["deform_kernel_ms"] for _ in range(7))
        out.append("%s %.3f" % (name, ts[3] * 1e3))
    print(tag, "kernel us (median of 7 x 1000 frames):", " | ".join(out), flush=True)
    ctx.close()
if "dense" in which:
    for name, V, B, M in (("c3", 30000, 200, 64), ("shard", 125184, 256, 64), ("c5", 1000000, 256, 64)):
        ctx = rz.DeformContext(0)
        mesh = synth.make_mesh_range(max(V, 30000), B, 0, V)
        ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
        d, mw = synth.make_morphs_dense_range(max(V, 30000), M, 0, V); ctx.upload_morphs_dense(d); del d
        ctx.set_pose(mesh["world"], mw)
        for _ in range(6):
            ctx.deform_n(300 if V < 500000 else 60); ctx.sync()
        n = 1000 if V < 500000 else 200
        ts = sorted((t["deform_kernel_ms"], t["frame_ms"]) for t in (ctx.time_frames(n) for _ in range(7)))
        print(tag, "%s kernel / frame us (median of 7 x %d frames): %.3f / %.3f" % (name, n, ts[3][0] * 1e3, ts[3][1] * 1e3), flush=True)
        ctx.close()
if "c4" in which:
    V, B, I = 30000, 200, 256
    ctx = rz.DeformContext(0)
    mesh = synth.make_mesh(V, B)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"]); ctx.upload_morphs_dense(None)
    ctx.set_instances(I)
    ctx.set_pose(np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)]))
    for _ in range(8):
        ctx.deform_n(300); ctx.sync()
    ts = sorted((t["deform_kernel_ms"], t["frame_ms"]) for t in (ctx.time_frames(500) for _ in range(9)))
    print(tag, "c4 kernel / frame us (median of 9 x 500 frames): %.3f / %.3f" % (ts[4][0] * 1e3, ts[4][1] * 1e3), flush=True)
    ctx.close()
if "anim" in which:
    for V, B, M, kind in ((30000, 200, 0, "none"), (28842, 349, 60, "sparse")):
        ctx = rz.DeformContext(0)
        mesh = synth.make_mesh(V, B)
        ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
        mw = None
        if kind == "sparse":
            off, vi, d3, mw = synth.make_morphs_demo_shape(V, M); ctx.upload_morphs_sparse(off, vi, d3)
        else:
            ctx.upload_morphs_dense(None)
        ctx.upload_skeleton_topology(mesh["parents"], mesh["bind"])
        rng = np.random.default_rng(1)
        nk = 8
        kq = rng.normal(size=(B, nk, 4)).astype(np.float32); kq /= np.linalg.norm(kq, axis=2, keepdims=True)
        extra = {}
        if M:
            extra = dict(mkey_off=np.arange(M + 1) * 2, mkey_frame=np.tile(np.array([0.0, 70.0], np.float32), M), mkey_weight=np.repeat(mw, 2),
                         feed_off=np.arange(M + 1), feed_track=np.arange(M), feed_ratio=np.ones(M, np.float32))
        ctx.upload_animation(np.arange(B), np.arange(B + 1) * nk, np.tile(np.arange(nk) * 10.0, B), kq, (rng.random((B, nk, 3), dtype=np.float32) - 0.5) * 0.2,
                             np.tile(np.array([20] * 8 + [107] * 8, np.uint8), B * nk), **extra)
        q = rng.normal(size=(B, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
        lt = ((rng.random((B, 3), dtype=np.float32) - 0.5) * 0.1).astype(np.float32)
        frames = np.stack([np.array([(3.0 + 0.5 * k) % 70], np.float32) for k in range(64)])
        res = {}
        for pose in ("world", "sampled", "local", "local+t"):
            if pose == "world":
                call, check = ctx.frame_call("world", mesh["world"], mw)
            elif pose == "sampled":
                call, check = ctx.frame_call("sampled", frames)
            else:
                call, check = ctx.frame_call("local", q, mw, lt if pose == "local+t" else None)
            for _ in range(300): call()
            ctx.sync()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(2000): call()
                ctx.sync(); best = min(best, (time.perf_counter() - t0) / 2000)
            check()
            rep_ms = sorted(ctx.time_frames(500)["frame_ms"] for _ in range(5))[2]
            res[pose] = (best * 1e6, rep_ms * 1e3)
        print(tag, "V=%d B=%d morphs=%s | per-frame loop / resident replay (us): " % (V, B, kind) +
              " | ".join("%s %.2f / %.2f" % (k, v[0], v[1]) for k, v in res.items()), flush=True)
        ctx.close()
