#!/usr/bin/env python3
"""Copies what tools/gpu_full.sh left under gpurun_out/full/ (scratch) into profiles/ (tracked) under the round's names, builds
profiles/<tag>_autotune_stability.json from the consecutive runs (usage: collect_profiles.py [tag], default r6), and regenerates DESIGN.md's tables (tools/design_tables.py)."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "full"), os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"
n = 0
for f in sorted(glob.glob(os.path.join(SRC, "bench_*.json"))):
    try:
        json.load(open(f))
    except Exception:       # noqa: BLE001
        print("skipping unreadable", f)
        continue
    dst = os.path.join(DST, "%s_%s" % (TAG, os.path.basename(f)))
    shutil.copy(f, dst)
    n += 1
    # `roofline.traffic` is a LOOKUP in profiles/pmc_traffic.json by (workload shape, kernel) — a stored record, not something the line
    # measures. A line printed before the round's PMC passes had run (the passes rename nothing, but a new kernel variant has no record
    # until they have) carries null: look it up now, in the records of the same session, and say so.
    try:
        d = json.load(open(dst))
        rf, c = d["roofline"], d["config"]
        tj = os.path.join(DST, "pmc_traffic.json")
        if rf.get("traffic") is None and os.path.exists(tj):
            sparse = {"demo": "_demo", "sparse2": "_sparse2"}.get(os.path.basename(f)[len("bench_"):-len(".json")], "")
            key = "V%d_B%d_M%d_I%d%s|%s" % (c["verts_per_gpu"], c["bones"], c["morphs"], c.get("instances_per_gpu", c["instances"]), sparse, rf["kernel"])
            rec = json.load(open(tj))
            if key in rec:
                rf["traffic"] = rec[key]["hbm_bytes_per_launch"]
                rf["traffic_source"] = "stored: profiles/pmc_traffic.json[%s] (%s) — looked up by tools/collect_profiles.py: the PMC passes ran after this line, in the same session" % (
                    key, rec[key].get("command", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"))
                json.dump(d, open(dst, "w"))
    except Exception as e:      # noqa: BLE001
        print("traffic lookup skipped for", f, e)
stab = {}
for name in ("c5", "c4"):
    runs = []
    for f in sorted(glob.glob(os.path.join(SRC, "stab_%s_*.json" % name))):
        try:
            d = json.load(open(f))
        except Exception:   # noqa: BLE001
            continue
        tab = d["config"].get("autotune_table") or []
        runs.append({"pick": d["config"].get("autotune_pick"), "kernel": d["roofline"]["kernel"], "grid": d["config"]["grid"], "morph_split": d["config"]["morph_split"],
                     "inst_group": d["config"].get("inst_group"), "ms_per_step": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms"],
                     "heuristic_ms": tab[0]["ms"] if tab else None, "best_candidate_ms": min(e["ms"] for e in tab) if tab else None})
    if runs:
        stab[{"c5": "C5 (python bench.py)", "c4": "C4 (--config c4)"}[name]] = runs
if stab:
    json.dump(stab, open(os.path.join(DST, "%s_autotune_stability.json" % TAG), "w"), indent=1)
for t in ("shard_scaling", "live_loop", "node_frame_bench", "pytest_gpu", "smoke", "plan_sweep", "parity", "pullbench", "fresh_plans", "fresh_plans_final", "onestep_sweep"):
    p = os.path.join(SRC, t + ".txt")
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(DST, "%s_%s.txt" % (TAG, t)))
print("copied %d bench lines" % n)
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "tools", "design_tables.py")]))
