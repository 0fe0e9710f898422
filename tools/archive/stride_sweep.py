#!/usr/bin/env python3
"""Round 6: is the non-monotonic roofline fraction by mesh size (NOTEBOOK R6.2: 625 k / 313 k / 156 k vertices are poor under every
launch shape) a property of the PLANE STRIDE (Vp * 4 bytes between the 6 + 3 M planes a lane streams in parallel)?

The library fixes Vp = V rounded up to 1 024, so the stride is swept by sweeping V itself in steps of 1 024 vertices around each base
size (the work changes by < 3 % over the sweep; `frac` normalises by the bytes). Every size is measured in a FRESH context (new
allocations), the sizes in shuffled order, the whole pass twice: a pattern that repeats in both passes belongs to the stride, one that
does not belongs to the placement (NOTEBOOK R5.3).

  python tools/archive/stride_sweep.py [--bases 625152,312576,156416] [--steps 24] [--step 1024] [--passes 2]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import reze_engine_amd as rz  # noqa: E402
from reze_engine_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", default="625152,312576,156416")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--step", type=int, default=1024)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--bones", type=int, default=256)
    ap.add_argument("--morphs", type=int, default=64)
    ap.add_argument("--tune", default="", help="k=v,... applied to every context (e.g. morph_split=2)")
    a = ap.parse_args()
    bases = [int(s) for s in a.bases.split(",")]
    sizes = sorted({b + k * a.step for b in bases for k in range(a.steps)})
    vmax = max(sizes)
    mesh = synth.make_mesh_range(vmax, a.bones, 0, vmax)
    deltas, mw = synth.make_morphs_dense_range(vmax, a.morphs, 0, vmax)
    rng = np.random.default_rng(5)
    res = {n: [] for n in sizes}
    for ps in range(a.passes):
        order = list(sizes)
        rng.shuffle(order)
        for n in order:
            ctx = rz.DeformContext(0)
            ctx.upload_mesh(*(np.ascontiguousarray(mesh[k][:n]) for k in ("pos", "nrm", "joints", "weights")))
            ctx.upload_skeleton(mesh["inv_bind"])
            ctx.upload_morphs_dense(np.ascontiguousarray(deltas[:, :n]))
            for kv in filter(None, a.tune.split(",")):
                k, v = kv.split("=")
                ctx.set_tuning(**{k: int(v)})
            ctx.set_pose(mesh["world"], mw)
            ctx.deform_n(300)
            ctx.sync()
            us = sorted(ctx.time_frames(a.frames)["deform_kernel_ms"] * 1e3 for _ in range(3))[1]
            byts = n * (60 + 12 * a.morphs) + a.bones * 128 + a.morphs * 4
            row = {"verts": n, "pass": ps, "stride_bytes": ((n + 1023) // 1024) * 4096, "split": ctx.get_tuning("effective_split"), "grid": ctx.get_tuning("effective_grid"),
                   "us": round(us, 3), "frac": round(byts / (us * 1e-6) / 8e12, 4)}
            res[n].append(row)
            print(json.dumps(row), flush=True)
            ctx.close()
    print("# verts  stride(B)  stride/4096 mod 64   frac per pass")
    for n in sizes:
        st = res[n][0]["stride_bytes"]
        print("# %8d %10d %6d   %s   split %d grid %d" % (n, st, (st // 4096) % 64, "  ".join("%.3f" % r["frac"] for r in res[n]), res[n][0]["split"], res[n][0]["grid"]))


if __name__ == "__main__":
    main()
