"""Per-frame loops of a HOST-animated crowd (C4: 256 x 30 000 verts / 200 bones), round 5: the pose pulled out of the pinned ring by
rz_pull_pose_kernel ("pose_pull" 1; world matrices as three rows per bone) against the runtime's copy ("pose_pull" 0), for world
matrices (rz_set_pose, 3.28 MB) and local rotations (rz_set_pose_local, 0.82 MB; the hierarchy solved in the skin kernel's front).
Per mode: the upload alone, the frame alone (resident replay), both per frame on one stream, both alternating between the context and a
fork (two frames in flight), and what ONE upload call costs the host thread when the GPU is idle (stream drained before every call)."""
import os, sys, time
if os.environ.get("RZ_TOOL_CPUS"):          # e.g. "0-63": run (and first-touch the pinned rings) on one NUMA node's cores
    lo, hi = os.environ["RZ_TOOL_CPUS"].split("-")
    os.sched_setaffinity(0, range(int(lo), int(hi) + 1))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth

V, B = 30000, 200
counts = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256]
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["world", "local"]
mesh = synth.make_mesh(V, B)
for I in counts:
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)]).astype(np.float32)
    rng = np.random.default_rng(4242)
    quats = rng.normal(size=(I, B, 4)).astype(np.float32)
    quats /= np.linalg.norm(quats, axis=2, keepdims=True)
    for kind in kinds:
        for pull in (0, 1):
            ctx = rz.DeformContext(0)
            ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"]); ctx.upload_morphs_dense(None)
            ctx.upload_skeleton_topology(mesh["parents"], mesh["bind"])
            ctx.set_instances(I)
            ctx.set_tuning(pose_pull=pull)
            fk = ctx.fork()
            L = ctx._L
            arr = rz.capi._f32(worlds if kind == "world" else quats).reshape(-1)
            ap = rz.capi._fptr(arr)
            if kind == "world":
                up = lambda h: L.rz_set_pose(h, ap, None)
            else:
                up = lambda h: L.rz_set_pose_local(h, ap, None, None)
            for x in (ctx, fk):
                up(x._h); x.deform_n(100); x.sync()
            pulled, rows = ctx.get_tuning("pose_pulled"), ctx.get_tuning("pose_rows")

            def loop(fn, n=1000, sync=lambda: (ctx.sync(), fk.sync())):
                for _ in range(100): fn()
                sync()
                best = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(n): fn()
                    sync(); best.append((time.perf_counter() - t0) / n)
                return sorted(best)[1] * 1e6
            h, h2 = ctx._h, fk._h
            t_up = loop(lambda: up(h))
            t_df = loop(lambda: L.rz_deform(h))
            t_both = loop(lambda: (up(h), L.rz_deform(h)))
            flip = [0]

            def two():
                flip[0] ^= 1
                x = h2 if flip[0] else h
                up(x); L.rz_deform(x)
            t_two = loop(two)
            host = 0.0
            for _ in range(300):
                ctx.sync()
                t0 = time.perf_counter(); up(h); host += time.perf_counter() - t0
            host = host / 300 * 1e6
            print("I=%d %s (%.2f MB handed over) pose_pull=%d [pulled %d, three rows %d]: upload alone %.1f us | frame alone %.1f us | upload + frame %.1f us | two in flight %.1f us | host cost of one upload call %.1f us" % (
                I, kind, arr.nbytes / 1e6, pull, pulled, rows, t_up, t_df, t_both, t_two, host), flush=True)
            fk.close(); ctx.close()
    t0 = time.perf_counter()
    dst = np.empty_like(worlds)
    for _ in range(300): np.copyto(dst, worlds)
    print("I=%d: a host memcpy of the world matrices (numpy) %.1f us; the link at 55 GB/s: %.1f us for 64 B/bone, %.1f us for 48 B/bone, %.1f us for 16 B/bone" % (
        I, (time.perf_counter() - t0) / 300 * 1e6, worlds.nbytes / 55e3, worlds.nbytes * 0.75 / 55e3, worlds.nbytes * 0.25 / 55e3), flush=True)
