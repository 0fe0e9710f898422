#!/usr/bin/env python3
"""Round 6: where does make_plan stand among the launch shapes at every shard size of C5 (N = 1, 2, 4, 8 and sizes between)?

For each shard size: the heuristic plan (morph_split = 0, grid_cap = 0) and every (morph split, grid) of the sweep are timed
round-robin — `rounds` rounds of `frames` event-timed frames each, the MEDIAN round counts — so that clock / placement drift
hits every candidate alike. Prints one JSON row per candidate, then per size the best shape and how far the heuristic is
from it. Exit status 1 when the heuristic is more than `--tol` (default 2 %) behind the best at any size.

  python tools/plan_sweep.py [--sizes 1000000,500224,250112,125184] [--splits 2,4,8] [--grids 128,192,245,...] [--rounds 5]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import reze_engine_amd as rz  # noqa: E402
from reze_engine_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1000000,500224,250112,125184")
    ap.add_argument("--splits", default="1,2,4,8")
    ap.add_argument("--grids", default="128,192,256,384,512,768,1024")
    ap.add_argument("--steps-per-wave", default="", help="instead of --grids: for every morph split S, the grids that give a wave k whole steps "
                                                        "(a step = 64 / S quads), k from this comma list")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--tol", type=float, default=0.02)
    ap.add_argument("--bones", type=int, default=256)
    ap.add_argument("--morphs", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "plan_sweep.json"))
    a = ap.parse_args()
    sizes = [int(s) for s in a.sizes.split(",")]
    splits = [int(s) for s in a.splits.split(",")]
    grids = [int(s) for s in a.grids.split(",")]
    vmax = max(sizes)
    mesh = synth.make_mesh_range(vmax, a.bones, 0, vmax)
    deltas, mw = synth.make_morphs_dense_range(vmax, a.morphs, 0, vmax)
    rows, worst = [], 0.0
    for n in sizes:
        ctx = rz.DeformContext(0)
        ctx.upload_mesh(*(np.ascontiguousarray(mesh[k][:n]) for k in ("pos", "nrm", "joints", "weights")))
        ctx.upload_skeleton(mesh["inv_bind"])
        ctx.upload_morphs_dense(np.ascontiguousarray(deltas[:, :n]))
        ctx.set_pose(mesh["world"], mw)
        ctx.deform_n(400)
        ctx.sync()
        if a.steps_per_wave:
            nq = (n + 3) // 4
            cands = [(0, 0)] + [(s, max(1, -(-nq // (4 * k * (64 // s))))) for s in splits for k in (int(x) for x in a.steps_per_wave.split(","))]
        else:
            cands = [(0, 0)] + [(s, g) for s in splits for g in grids]
        seen, uniq = {}, []
        for s, g in cands:
            ctx.set_tuning(morph_split=s, grid_cap=g)
            key = (ctx.get_tuning("effective_split"), ctx.get_tuning("effective_grid"), ctx.get_tuning("effective_out_cap"))
            if (s, g) != (0, 0) and key in seen:
                continue
            seen.setdefault(key, (s, g))
            uniq.append((s, g, key))
        t = {c: [] for c in uniq}
        for r in range(-1, a.rounds):
            for c in uniq:
                ctx.set_tuning(morph_split=c[0], grid_cap=c[1])
                tm = ctx.time_frames(20 if r < 0 else a.frames)
                if r >= 0:
                    t[c].append(tm["frame_ms"] * 1e3)
        alg = tm["algorithmic_bytes_per_frame"]
        med = {c: float(np.median(v)) for c, v in t.items()}
        best = min(med, key=med.get)
        heur = uniq[0]
        for c in uniq:
            row = {"verts": n, "request": {"morph_split": c[0], "grid_cap": c[1]}, "split": c[2][0], "grid": c[2][1], "out_cap": c[2][2],
                   "us": round(med[c], 3), "us_min": round(min(t[c]), 3), "us_max": round(max(t[c]), 3),
                   "frac": round(alg / med[c] / 1e3 / 8000.0, 4), "heuristic": c is heur, "best": c is best}
            rows.append(row)
            print(json.dumps(row), flush=True)
        gap = med[heur] / med[best] - 1.0
        worst = max(worst, gap)
        print(json.dumps({"verts": n, "heuristic": {"split": heur[2][0], "grid": heur[2][1], "us": round(med[heur], 3)},
                          "best": {"split": best[2][0], "grid": best[2][1], "us": round(med[best], 3)},
                          "heuristic_behind_best_pct": round(100 * gap, 2)}), flush=True)
        ctx.close()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rows, open(a.out, "w"), indent=1)
    if worst > a.tol:
        print("FAIL: the heuristic plan is %.1f %% behind the best shape somewhere (tolerance %.0f %%)" % (100 * worst, 100 * a.tol))
        sys.exit(1)


if __name__ == "__main__":
    main()
