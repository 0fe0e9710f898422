#!/usr/bin/env python3
"""Kernel-variant sweep on one MI355X (run through gpurun; needs `make -C reze-engine_amd/csrc variants`). Times the fused morph+skin kernel
alone with HIP events (rz_time_frames) for every tuning combination on the BASELINE configs and
prints achieved algorithmic GB/s. Output: a table on stdout + gpurun_out/sweep.json."""
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import reze_engine_amd as rz  # noqa: E402
from reze_engine_amd import synth  # noqa: E402


def setup(ctx, V, B, M, I=1):
    mesh = synth.make_mesh(V, B)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    ctx.upload_skeleton(mesh["inv_bind"])
    mw = None
    if M:
        deltas, mw = synth.make_morphs_dense(V, M)
        ctx.upload_morphs_dense(deltas)
        del deltas
    else:
        ctx.upload_morphs_dense(None)
    ctx.set_instances(I)
    worlds = mesh["world"]
    if I > 1:
        worlds = np.stack([synth.make_pose(mesh["parents"], mesh["bind"], B, seed=1000 + i) for i in range(I)])
        if mw is not None:
            mw = np.tile(mw, (I, 1))
    ctx.set_pose(worlds, mw)


def setup_sparse(ctx, V, B, M):
    mesh = synth.make_mesh(V, B)
    ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"])
    ctx.upload_skeleton(mesh["inv_bind"])
    off, idx, d3, mw = synth.make_morphs_sparse(V, M)
    ctx.upload_morphs_sparse(off, idx, d3)
    ctx.set_instances(1)
    ctx.set_pose(mesh["world"], mw)


def run(ctx, name, frames, grid):
    rows = []
    keys = list(grid.keys())
    for combo in itertools.product(*[grid[k] for k in keys]):
        kw = dict(zip(keys, combo))
        try:
            ctx.set_tuning(**kw)
            ctx.time_frames(5)
            best = None
            for _ in range(3):
                t = ctx.time_frames(frames)
                if best is None or t["deform_kernel_ms"] < best["deform_kernel_ms"]:
                    best = t
            gbps = best["algorithmic_bytes_per_frame"] / (best["deform_kernel_ms"] * 1e-3) / 1e9
            row = dict(config=name, **kw, kernel_ms=best["deform_kernel_ms"], frame_ms=best["frame_ms"],
                       prep_ms=best["prep_kernel_ms"], gbps=gbps, frac=gbps / 8000.0,
                       gverts=best["verts_per_frame"] / (best["deform_kernel_ms"] * 1e-3) / 1e9,
                       S=ctx.get_tuning("effective_split"), U=ctx.get_tuning("effective_unroll"),
                       F=ctx.get_tuning("effective_fast"), grid=ctx.get_tuning("effective_grid"))
        except Exception as e:   # keep sweeping
            row = dict(config=name, **kw, error=str(e))
        rows.append(row)
        print(json.dumps(row), flush=True)
    return rows


def main():
    which = sys.argv[1:] or ["c5", "c5shard", "c4", "c3", "c2", "real"]
    # the sweep covers variants the product does not carry (unroll = 4, geo_lds = 1, ...): it runs on the all-variants build
    ctx = rz.DeformContext(0, lib=rz.capi.load(rz.capi.VARIANTS_LIB_PATH))
    out = []
    t0 = time.time()
    base = dict(morph_split=[0], unroll=[0], nontemporal=[1], nt_store=[-1], geo_lds=[0], grid_cap=[0], fast=[-1], inst_loop=[-1])

    def g(**kw):
        d = dict(base)
        d.update(kw)
        return d
    if "c5" in which:
        setup(ctx, 1000000, 256, 64)
        out += run(ctx, "C5 1M/256/64", 40, g(morph_split=[1, 2, 4], unroll=[4, 8], grid_cap=[256, 512, 768], nt_store=[0, 1]))
    if "c5shard" in which:
        for nr in (8, 4, 2):
            b, n = rz.shard_range(1000000, nr, 0)
            setup(ctx, n, 256, 64)
            out += run(ctx, "C5 shard 1/%d (%d)" % (nr, n), 300 if nr == 8 else 100,
                       g(morph_split=[1, 2, 4, 8], unroll=[4, 8], grid_cap=[256, 512, 1024], nt_store=[0, 1]))
    if "c4" in which:
        setup(ctx, 30000, 200, 0, I=256)
        out += run(ctx, "C4 256x30k pose-loop", 100, g(fast=[-1, 0], inst_loop=[4, 8], grid_cap=[512, 768, 1024, 2048]))
        out += run(ctx, "C4 256x30k generic", 100, g(nt_store=[1], inst_loop=[0], grid_cap=[1024, 2048, 4096]))
    if "c3" in which:
        setup(ctx, 30000, 200, 64)
        out += run(ctx, "C3 30k/200/64", 300, g(morph_split=[1, 2, 4, 8], unroll=[4, 8], fast=[1, 0], grid_cap=[256, 512]))
    if "c2" in which:
        setup(ctx, 30000, 200, 0)
        out += run(ctx, "C2 30k/200/0", 300, g(geo_lds=[1, 0], fast=[1, 0], grid_cap=[64, 128, 256]))
    if "real" in which:
        setup_sparse(ctx, 28842, 349, 60)
        out += run(ctx, "demo-shaped 28842/349/60 sparse", 300, g(geo_lds=[1, 0], fast=[1, 0]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sweep.json"), "w"), indent=1)
    print("sweep done in %.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
