"""Device-animated single character: three kernels per frame (rz_fk_kernel [+ rz_prep_kernel] + deform) vs ONE (fuse_fk: every
workgroup of the deform kernel solves the hierarchy itself). Per-frame loops through the raw C ABI, GPU-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
for V, B, M, kind in ((30000, 200, 30, "sparse"), (30000, 200, 0, "none"), (30000, 200, 64, "dense"), (125184, 256, 64, "dense"), (1000000, 256, 64, "dense")):
    ctx = rz.DeformContext(0)
    mesh = synth.make_mesh_range(max(V, 30000), B, 0, V)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"])
    mw = None
    if kind == "dense":
        d, mw = synth.make_morphs_dense_range(max(V, 30000), M, 0, V); ctx.upload_morphs_dense(d)
    elif kind == "sparse":
        off, vi, d3, mw = synth.make_morphs_sparse(V, M); ctx.upload_morphs_sparse(off, vi, d3)
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
    frames = np.stack([np.array([(3.0 + 0.5 * k) % 70], np.float32) for k in range(64)])
    res = {}
    for pose in ("sampled", "local"):
        for fuse in (0, 1):
            ctx.set_tuning(fuse_fk=fuse)
            call, check = ctx.frame_call("sampled", frames) if pose == "sampled" else ctx.frame_call("local", q, mw)
            for _ in range(300): call()
            ctx.sync()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(1500): call()
                ctx.sync(); best = min(best, (time.perf_counter() - t0) / 1500)
            check()
            res[(pose, fuse)] = best * 1e6
    print("V=%7d B=%d morphs=%-6s | sampled pose: 3 kernels %.2f us -> fused %.2f us | local pose: 3 kernels %.2f us -> fused %.2f us" % (
        V, B, kind, res[("sampled", 0)], res[("sampled", 1)], res[("local", 0)], res[("local", 1)]), flush=True)
    ctx.close()
