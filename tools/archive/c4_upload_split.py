"""Where does a host-animated crowd's per-frame time go? C4 (256 x 30 000 / 200 bones): rz_set_pose alone, rz_deform alone,
both per frame (one stream), both alternating between the context and a fork (two in flight) — and the same at 128 / 64 poses."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
V, B = 30000, 200
mesh = synth.make_mesh(V, B)
for I in (256, 128, 64):
    ctx = rz.DeformContext(0)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"]); ctx.upload_morphs_dense(None)
    ctx.set_instances(I)
    worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)])
    L, h = ctx._L, ctx._h
    wp = rz.capi._fptr(rz.capi._f32(worlds).reshape(-1))
    ctx.set_pose(worlds); ctx.deform_n(200); ctx.sync()

    def loop(fn, n=1500):
        for _ in range(100): fn()
        ctx.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n): fn()
            ctx.sync(); best = min(best, (time.perf_counter() - t0) / n)
        return best * 1e6
    t_up = loop(lambda: L.rz_set_pose(h, wp, None))
    t_df = loop(lambda: L.rz_deform(h))
    t_both = loop(lambda: (L.rz_set_pose(h, wp, None), L.rz_deform(h)))
    t0 = time.perf_counter()
    for _ in range(300): np.copyto(np.empty_like(worlds), worlds)
    t_cp = (time.perf_counter() - t0) / 300 * 1e6
    print("I=%d (%.2f MB of matrices): set_pose alone %.1f us | deform alone %.1f us | both %.1f us | a host memcpy of the matrices (numpy, incl. allocation) %.1f us | PCIe at 55 GB/s would be %.1f us" % (
        I, worlds.nbytes / 1e6, t_up, t_df, t_both, t_cp, worlds.nbytes / 55e3), flush=True)
    ctx.close()
