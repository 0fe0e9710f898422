#!/usr/bin/env python3
"""Round 6: the persistent grid (two workgroups per CU, every wave an equal run) against ONE STEP PER WAVE on as many workgroups as that
takes (the hardware deals them out as slots free up), at mesh sizes between and at the shard sizes of N = 1 ... 8. Every size in a fresh
context; heuristic plan, S = 4 / 2 / 8 with one step per wave, timed round-robin (3 rounds of `frames` frames, median).
  python tools/onestep_sweep.py [--sizes ...]"""
import argparse, json, os, sys      # noqa: E401
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import reze_engine_amd as rz  # noqa: E402
from reze_engine_amd import synth  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="")
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--bones", type=int, default=256)
    ap.add_argument("--morphs", type=int, default=64)
    a = ap.parse_args()
    if a.sizes:
        sizes = [int(s) for s in a.sizes.split(",")]
    else:
        sizes, n = [], 110000.0
        while n < 1.05e6:
            sizes.append(int(n) // 256 * 256)
            n *= 1.06
        sizes += [1000000, 500224, 333568, 250112, 200192, 166912, 143104, 125184]
        sizes = sorted(set(sizes))
    vmax = max(sizes)
    mesh = synth.make_mesh_range(vmax, a.bones, 0, vmax)
    deltas, mw = synth.make_morphs_dense_range(vmax, a.morphs, 0, vmax)
    print("# verts | heuristic S/grid steps-per-wave us frac | S4 one step: grid us (vs heur %) | S2 one step | S8 one step")
    for n in sizes:
        ctx = rz.DeformContext(0)
        ctx.upload_mesh(*(np.ascontiguousarray(mesh[k][:n]) for k in ("pos", "nrm", "joints", "weights")))
        ctx.upload_skeleton(mesh["inv_bind"])
        ctx.upload_morphs_dense(np.ascontiguousarray(deltas[:, :n]))
        ctx.set_pose(mesh["world"], mw)
        ctx.deform_n(300)
        ctx.sync()
        nq = (n + 3) // 4
        plans = [(0, 0)] + [(s, -(-nq // (4 * (64 // s)))) for s in (4, 2, 8)]
        t = {p: [] for p in plans}
        shape = {}
        for r in range(3):
            for p in plans:
                ctx.set_tuning(morph_split=p[0], grid_cap=p[1])
                shape[p] = (ctx.get_tuning("effective_split"), ctx.get_tuning("effective_grid"))
                t[p].append(ctx.time_frames(a.frames)["deform_kernel_ms"] * 1e3)
        byts = n * (60 + 12 * a.morphs) + a.bones * 128 + a.morphs * 4
        med = {p: float(np.median(t[p])) for p in plans}
        h = plans[0]
        spw = nq / (64 // shape[h][0]) / (shape[h][1] * 4)
        row = {"verts": n, "heuristic": {"split": shape[h][0], "grid": shape[h][1], "steps_per_wave": round(spw, 3), "us": round(med[h], 3), "frac": round(byts / med[h] / 8e6, 4)}}
        for p in plans[1:]:
            row["S%d_one_step" % p[0]] = {"grid": shape[p][1], "us": round(med[p], 3), "vs_heuristic_pct": round((med[p] / med[h] - 1) * 100, 2)}
        print(json.dumps(row), flush=True)
        ctx.close()

if __name__ == "__main__":
    main()
