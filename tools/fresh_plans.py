#!/usr/bin/env python3
"""Round 6: is a launch shape robust, or only good where the allocator happened to put the buffers? Every (size, generator) runs in a
FRESH process — new context, new allocations — and times a handful of plans round-robin; repeated, the spread of one plan across the
processes is its placement sensitivity (250 112 vertices, S = 4 / 489 workgroups: 32.4 - 36.6 us; S = 4 / 245: 34.4 - 34.9 us).
  python tools/fresh_plans.py [sizes] [repetitions] [plans: S:G,... with G = x1 for one step per wave]   -> profiles/r6_fresh_plans.txt"""
import json, os, subprocess, sys       # noqa: E401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
import reze_engine_amd as rz
from reze_engine_amd import synth
n = int(sys.argv[1]); gen = sys.argv[2]
if gen == "range":
    mesh = synth.make_mesh_range(n, 256, 0, n); deltas, mw = synth.make_morphs_dense_range(n, 64, 0, n)
else:
    mesh = synth.make_mesh(n, 256); deltas, mw = synth.make_morphs_dense(n, 64)
ctx = rz.DeformContext(0)
ctx.upload_mesh(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"]); ctx.upload_skeleton(mesh["inv_bind"]); ctx.upload_morphs_dense(deltas)
ctx.set_pose(mesh["world"], mw)
for _ in range(20): ctx.deform_n(200)
ctx.sync()
plans = [(0, 0), (4, 256), (4, 512), (2, 256), (4, 1024)]
if len(sys.argv) > 3:       # "S:G,..." — G = workgroups, or x1 = as many as one step per wave takes
    nq = (n + 3) // 4
    plans = [(0, 0)] + [(int(s), -(-nq // (4 * (64 // int(s)))) if g == "x1" else int(g)) for s, g in (x.split(":") for x in sys.argv[3].split(","))]
t = {p: [] for p in plans}
for r in range(4):
    for p in plans:
        ctx.set_tuning(morph_split=p[0], grid_cap=p[1])
        tm = ctx.time_frames(200)
        if r: t[p].append(tm["frame_ms"] * 1e3)
out = {}
for p in plans:
    ctx.set_tuning(morph_split=p[0], grid_cap=p[1])
    out["S%%d/%%d" %% (ctx.get_tuning("effective_split"), ctx.get_tuning("effective_grid")) + ("(heur)" if p == (0, 0) else "")] = round(float(np.median(t[p])), 2)
print(json.dumps({"verts": n, "gen": gen, "us": out}))
''' % ROOT
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [250112, 218880, 187648, 281600]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = [sys.argv[3]] if len(sys.argv) > 3 else []
for n in sizes:
    for rep in range(reps):
        for gen in ("range", "whole"):
            p = subprocess.run([sys.executable, "-c", CHILD, str(n), gen] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
            print(lines[-1] if lines else "FAILED " + p.stderr.decode()[-300:], flush=True)
