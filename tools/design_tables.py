#!/usr/bin/env python3
"""Regenerates the result tables of DESIGN.md from the tracked evidence under profiles/ — so that no number in them is typed.

    python tools/design_tables.py            rewrite the block between the GENERATED markers of DESIGN.md
    python tools/design_tables.py --check    exit 1 when DESIGN.md's block differs from what profiles/ says (tests/test_docs.py)

Sources (all written on the GPU box by tools/gpu_full.sh / tools/gpu_profile.sh -> tools/parse_prof.py, then copied to profiles/):
    profiles/r6_bench_*.json          bench.py lines (the driver contract), one per workload
    profiles/r6_kernel_stats_*.txt    rocprofv3 --kernel-trace --stats of the same bench commands
    profiles/pmc_traffic.json         HBM bytes per launch from the PMC passes, keyed "<shape>|<kernel>"
    profiles/r6_autotune_stability.json   what the launch-shape search picked in consecutive runs
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
TAG = "r6"
PREV = "r5"
BEGIN = "<!-- BEGIN GENERATED (tools/design_tables.py — do not edit by hand) -->"
END = "<!-- END GENERATED -->"

# file suffix -> what the line is (order = table order)
LINES = [
    ("c5", "C5: 1 M verts / 256 bones / 64 dense morphs, 1 GPU — `python bench.py` (the driver's line)"),
    ("c5_steps20", "the same with the driver's flags — `--steps 20 --warmup 5`"),
    ("shard2", "one 1/2 shard of C5 (500 224 verts) — `--verts 500224`"),
    ("shard4", "one 1/4 shard of C5 (250 112 verts) — `--verts 250112`"),
    ("shard8", "one 1/8 shard of C5 (125 184 verts), one stream — `--verts 125184 --frames-in-flight 1`"),
    ("shard8_steps20", "the same shard with the driver's flags — `--verts 125184 --steps 20 --warmup 5`"),
    ("shard8_auto", "the same shard, default `--frames-in-flight auto`"),
    ("c4", "C4: 256 instances x 30 000 verts / 200 bones — `--config c4`"),
    ("c4_devicefk", "C4 with the hierarchy solved on the GPU — `--config c4 --device-fk`"),
    ("c4_sampled", "C4 with motion sampling + hierarchy on the GPU — `--config c4 --device-fk --device-sampling`"),
    ("c3", "C3: 30 000 verts / 200 bones / 64 dense morphs — `--config c3`"),
    ("c2", "C2: 30 000 verts / 200 bones / no morphs — `--config c2`"),
    ("demo", "demo-shaped: 28 842 verts / 349 bones / 60 sparse morphs, 36 397 offsets on one face region — `--config demo`"),
    ("sparse2", "the same mesh, sparse morphs spread at 2 % density — `--config sparse2`"),
    ("c5_allgather1", "C5 through torch.distributed with one rank + the RCCL all-gather — `--allgather`"),
    ("rehearse8", "8 ranks sharing ONE GPU over gloo (plumbing rehearsal, not a scaling number) — `--gpus 8 --share-gpu --dist-backend gloo`"),
    ("c4_rehearse8", "C4 over 8 ranks sharing ONE GPU, sharded along the instance axis (plumbing rehearsal) — `--config c4 --gpus 8 --share-gpu --dist-backend gloo`"),
]
STATS = {"c5": "c5", "shard2": "shard2", "shard4": "shard4", "shard8": "shard", "c4": "c4", "c3": "c3", "demo": "demo"}


def load(suffix, tag=None):
    p = os.path.join(PROF, "%s_bench_%s.json" % (tag or TAG, suffix))
    if not os.path.exists(p):
        return None
    try:
        return json.load(open(p))
    except Exception:       # noqa: BLE001
        return None


def rocprof_row(stats_name, kernel):
    """(calls, avg_us, min_us, max_us[, workgroups]) of `kernel` in profiles/r6_kernel_stats_<name>.txt: its most-launched launch
    shape when the file has the per-shape section (the plan the bench loops ran), else the all-shapes row of the stats"""
    p = os.path.join(PROF, "%s_kernel_stats_%s.txt" % (TAG, stats_name))
    if not os.path.exists(p):
        return None
    best = None
    for ln in open(p):
        if ln.startswith("SHAPE ") and kernel in ln:
            f = ln[len("SHAPE "):].rsplit(None, 7)
            row = (int(f[4]), float(f[5]), float(f[6]), float(f[7]), int(f[1]))
            if best is None or row[0] > best[0]:
                best = row
    if best is not None:
        return best
    for ln in open(p):
        if kernel in ln:
            m = re.search(r"\)?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s*$", ln)
            if m:
                return int(m.group(1)), float(m.group(3)), float(m.group(4)), float(m.group(5))
    return None


def us(ms):
    return "—" if ms is None else "%.2f" % (ms * 1e3)


def build():
    out = [BEGIN, ""]
    out.append("**Tracked bench lines** (`profiles/%s_bench_*.json`; times in µs; `frac` = algorithmic bytes ÷ event-timed kernel ÷ 8 TB/s, "
               "`frame` = the same over the whole frame; rocprof = average of that kernel in `profiles/%s_kernel_stats_*.txt`):" % (TAG, TAG))
    out.append("")
    out.append("| line | N | step (event span) | step (host wall) | verts/s | in flight | kernel the plan launches | kernel (events) | rocprof avg | frac | frame | traffic ÷ algorithmic | + pose upload | sampled on GPU |")
    out.append("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    have = []
    for suffix, what in LINES:
        d = load(suffix)
        if d is None:
            continue
        have.append(suffix)
        c, r = d["config"], d["roofline"]
        rp = rocprof_row(STATS[suffix], r["kernel"]) if suffix in STATS else None
        tr = "—" if not r.get("traffic") else "%.4f" % (r["traffic"] / r["algorithmic_bytes_per_launch"])
        out.append("| %s | %d | %s | %s | %.4g | %s | `%s` | %s | %s | %.3f | %.3f | %s | %s | %s |" % (
            what, d["n_gpus"], us(d["ms_per_step"]), us(c.get("ms_per_step_host_wall")), d["value"], c.get("frames_in_flight", 1), r["kernel"], us(r["kernel_ms"]),
            "—" if rp is None else "%.2f" % rp[1], r["frac"], r["frame_frac"], tr, us(c.get("frame_ms_with_pose_upload")), us(c.get("frame_ms_device_sampled_pose"))))
    out.append("")
    # the scaling projection: rank 0's shard at N = 1, 2, 4, 8 on ONE GPU, by the event span and by the host's clock, at 500 and at the driver's 20 steps
    proj = [(1, "c5"), (2, "shard2"), (4, "shard4"), (8, "shard8_auto")]
    if all(load(sfx) is not None for _, sfx in proj):
        out.append("**Projected strong scaling of C5 from rank 0's shard on ONE GPU** (`ms_per_step_one_stream` of the lines above: the hipEvent span of the K steps; "
                   "the 8-GPU run itself is the driver's; `host wall` = the host's clock around the same K steps; the 20-step rows are the driver's flags):")
        out.append("")
        out.append("| N | shard | steps | step, event span (µs) | projected speed-up | step, host wall (µs) | speed-up by the host wall | fixed host cost per timed region (µs) | frac |")
        out.append("|---|---|---|---|---|---|---|---|---|")
        for steps_sfx in ("", "_steps20"):
            base = load("c5" + steps_sfx)
            if base is None:
                continue
            for n, sfx in proj:
                d = load((sfx.replace("_auto", "") if steps_sfx else sfx) + steps_sfx)
                if d is None:
                    continue
                c, cb = d["config"], base["config"]
                out.append("| %d | %d | %d | %s | %.2f | %s | %.2f | %.1f | %.3f |" % (
                    n, c["verts_per_gpu"], d["steps"], us(c["ms_per_step_one_stream"]), cb["ms_per_step_one_stream"] / c["ms_per_step_one_stream"],
                    us(c.get("ms_per_step_one_stream_host_wall")), cb["ms_per_step_one_stream_host_wall"] / c["ms_per_step_one_stream_host_wall"],
                    c.get("host_fixed_cost_us_per_timed_region", float("nan")), d["roofline"]["frac"]))
        out.append("")
    d4 = load("c4")
    if d4 is not None and d4["config"].get("frame_ms_with_pose_mapped") is not None:
        c = d4["config"]
        out.append("**A host-animated crowd's per-frame loop (C4, world matrices; `profiles/%s_bench_c4.json`)** — resident replay %s µs; `rz_set_pose` + `rz_deform` per frame %s µs (two in flight %s); "
                   "caller-written poses (`rz_map_pose` rows + one memmove standing in for the pose solve + `rz_commit_pose` + `rz_deform`) %s µs (two in flight %s); the same without the write "
                   "(what the library and the host link cost) %s µs; local rotations and sampled motion: the `--device-fk` lines." % (
                       TAG, us(c["ms_per_step_one_stream"]), us(c["frame_ms_with_pose_upload"]), us(c.get("frame_ms_with_pose_upload_two_in_flight")), us(c["frame_ms_with_pose_mapped"]),
                       us(c.get("frame_ms_with_pose_mapped_two_in_flight")), us(c.get("frame_ms_with_pose_mapped_protocol_only"))))
        out.append("")
    # what the round changed: the same lines of the previous round's tracked evidence
    rows = []
    for suffix, what in LINES:
        a, b = load(suffix, PREV), load(suffix)
        if a is None or b is None or suffix in ("rehearse8", "c5_allgather1", "c4_rehearse8"):
            continue
        ca, cb = a["config"], b["config"]

        def pair(x, y):
            return "—" if x is None or y is None else "%.2f → %.2f" % (x * 1e3, y * 1e3)
        rows.append("| %s | %s | %s | %s | %s |" % (suffix, pair(ca.get("ms_per_step_one_stream"), cb.get("ms_per_step_one_stream")), pair(a["roofline"]["kernel_ms"], b["roofline"]["kernel_ms"]),
                                                 pair(ca.get("frame_ms_with_pose_upload"), cb.get("frame_ms_with_pose_upload")), pair(ca.get("frame_ms_device_sampled_pose"), cb.get("frame_ms_device_sampled_pose"))))
    if rows:
        out.append("**Round %s → round %s, line by line** (`profiles/%s_bench_*.json` against `profiles/%s_bench_*.json`, µs; different boxes: differences under ≈ 3 %% are box noise "
                   "(`profiles/r4_box_variance.txt`); the same-session A/B runs are in `profiles/%s_ab_*.txt`; both rounds' lines run on the cores of the GPU's NUMA node; round 6's steps are event spans, round 5's the host's clock):" % (PREV[1:], TAG[1:], PREV, TAG, TAG))
        out.append("")
        out.append("| line | frame, one stream | kernel (events) | frame + pose upload | frame, pose sampled on the GPU |")
        out.append("|---|---|---|---|---|")
        out += rows
        out.append("")
    nb = []
    for tag in (PREV, TAG):
        pth = os.path.join(PROF, "%s_node_frame_bench.txt" % tag)
        if os.path.exists(pth):
            for ln in open(pth):
                if ln.startswith("{"):
                    try:
                        nb.append((tag, json.loads(ln)))
                    except Exception:       # noqa: BLE001
                        pass
    if len(nb) == 2:
        out.append("**Through Node** (`profiles/*_node_frame_bench.txt`: Node → N-API → C ABI → MI355X, the same PMX + VMD; `gpuFrameUs` = resident replay of the last frame, `usPerFrame` = the per-frame loop incl. the JavaScript side):")
        out.append("")
        out.append("| mode | round %s: µs per frame / GPU frame | round %s: µs per frame / GPU frame |" % (PREV[1:], TAG[1:]))
        out.append("|---|---|---|")
        for k in ("host", "deviceFK", "sampled", "deviceFK2", "sampled2"):
            if k in nb[0][1] and k in nb[1][1]:
                out.append("| %s | %.2f / %.2f | %.2f / %.2f |" % (k, nb[0][1][k]["usPerFrame"], nb[0][1][k]["gpuFrameUs"], nb[1][1][k]["usPerFrame"], nb[1][1][k]["gpuFrameUs"]))
        out.append("")
    # the profiled runs themselves: events vs rocprof in the SAME run
    rows = []
    for name in ("c5", "shard2", "shard4", "shard", "c4", "c3", "demo"):
        pj = os.path.join(PROF, "%s_bench_under_rocprof_%s.json" % (TAG, name))
        if not os.path.exists(pj):
            continue
        try:
            d = json.load(open(pj))
        except Exception:       # noqa: BLE001
            continue
        rp = rocprof_row(name, d["roofline"]["kernel"])
        if rp is None:
            continue
        rows.append("| %s | `%s` | %s | %s | %.2f | %.2f | %d | %.3f |" % (name, d["roofline"]["kernel"], us(d["ms_per_step"]), us(d["roofline"]["kernel_ms"]), rp[1], rp[2], rp[0],
                                                                  rp[1] / (d["roofline"]["kernel_ms"] * 1e3)))
    if rows:
        out.append("**The same numbers inside ONE run** (`profiles/%s_bench_under_rocprof_*.json` = the bench line printed under rocprofv3, against the rocprofv3 average of "
                   "the most-launched shape of that kernel in the same run, `profiles/%s_kernel_stats_*.txt`; profiled runs clock lower than un-profiled ones):" % (TAG, TAG))
        out.append("")
        out.append("| workload | kernel | step (µs) | kernel by events (µs) | rocprof avg (µs) | rocprof min (µs) | launches | rocprof ÷ events |")
        out.append("|---|---|---|---|---|---|---|---|")
        out += rows
        out.append("")
    # one stream vs two frames in flight, and what the search did
    out.append("**Both frame modes of every line** (the scaling ratio must be read mode for mode) **and the launch-shape search:**")
    out.append("")
    out.append("| line | one stream | two frames in flight | chosen | search: heuristic plan | picked entry | its time | + pose upload, two in flight |")
    out.append("|---|---|---|---|---|---|---|---|")
    for suffix in have:
        d = load(suffix)
        c = d["config"]
        tab, pick = c.get("autotune_table"), c.get("autotune_pick")
        h = "—" if not tab else "%.2f" % (tab[0]["ms"] * 1e3)
        pk = "—" if not tab else ("entry %d%s" % (pick, " (the heuristics)" if pick == 0 else " (split %d, cap %d, poses %d)" % (tab[pick]["morph_split"], tab[pick]["grid_cap"], tab[pick]["inst_loop"])))
        pt = "—" if not tab else "%.2f" % (tab[pick]["ms"] * 1e3)
        out.append("| %s | %s | %s | %d | %s | %s | %s | %s |" % (suffix, us(c.get("ms_per_step_one_stream")), us(c.get("ms_per_step_two_frames_in_flight")),
                                                           c.get("frames_in_flight", 1), h, pk, pt, us(c.get("frame_ms_with_pose_upload_two_in_flight"))))
    out.append("")
    # CPU baseline + RCCL evidence
    d = load("c5")
    if d and d.get("cpu_baseline"):
        cb = d["cpu_baseline"]
        out.append("**CPU baseline of the driver line** (`cpu_baseline`, reported, not a target): %.3g %s on %d threads — %s." % (cb["value"], cb["unit"], cb["cores"], cb["sample"]))
        out.append("")
    d = load("c5_allgather1")
    if d:
        rc = d["config"]["ranks"][0].get("rccl") or {}
        out.append("**RCCL evidence at N = 1** (`%s_bench_c5_allgather1.json`): communicator count %s, user rank %s, `%s` version %s (reused from the process: %s); all-gather %s µs per call, outside `value`." % (
            TAG, rc.get("comm_count"), rc.get("comm_user_rank"), rc.get("path"), rc.get("version"), rc.get("reused"), us(d["config"].get("allgather_ms"))))
        out.append("")
    d = load("c4_rehearse8")
    if d:
        ranks = d["config"]["ranks"]
        out.append("**C4 over 8 ranks on one GPU, instance-sharded** (`%s_bench_c4_rehearse8.json`): %s; per rank (first instance, instances): %s; no communicator (%s)." % (
            TAG, d["config"]["workload"], ", ".join("(%d, %d)" % (r["instance_begin"], r["instances"]) for r in ranks), (ranks[0].get("rccl") or {}).get("skipped")))
        out.append("")
    d = load("rehearse8")
    if d:
        ranks = d["config"]["ranks"]
        plans = sorted({(r["kernel"], r["morph_split"]) for r in ranks})
        out.append("**8-rank rehearsal on one GPU** (`%s_bench_rehearse8.json`; launched by: %s): %d ranks, %d distinct plan(s) %s, kernel %.2f–%.2f µs over the ranks "
                   "(all eight shards take turns on the one GPU, so the step time says nothing about scaling)." % (
                       TAG, d["config"].get("launched_by"), len(ranks), len(plans), ", ".join("`%s`" % p[0] for p in plans),
                       d["config"]["kernel_ms_min_over_ranks"] * 1e3, d["config"]["kernel_ms_max_over_ranks"] * 1e3))
        out.append("")
    # PMC traffic
    tj = os.path.join(PROF, "pmc_traffic.json")
    if os.path.exists(tj):
        rec = json.load(open(tj))
        out.append("**HBM traffic per launch from the PMC counters** (`profiles/pmc_traffic.json`: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, "
                   "calibrated in the same run on `tools/membench`'s known-byte kernels):")
        out.append("")
        out.append("| workload shape | kernel | launches averaged | measured bytes | algorithmic bytes | ratio |")
        out.append("|---|---|---|---|---|---|")
        for k in sorted(rec):
            if k.startswith("_") or "|" not in k:
                continue
            r = rec[k]
            alg = None if r.get("algorithmic_read_bytes") is None else r["algorithmic_read_bytes"] + r["algorithmic_write_bytes"]
            out.append("| %s | `%s` | %d | %.4g | %s | %s |" % (k.split("|")[0], r["kernel"], r["launches_counted"], r["hbm_bytes_per_launch"],
                                                            "—" if alg is None else "%.4g" % alg, "—" if r.get("traffic_over_algorithmic") is None else "%.4f" % r["traffic_over_algorithmic"]))
        cal = rec.get("_calibration", {})
        if cal:
            out.append("")
            out.append("Calibration factors of that run: FETCH_SIZE × %.5f (16 B/lane nontemporal reads), WRITE_SIZE × %.4f (nontemporal 12 B/lane stores) / × %.4f (plain)." % (
                cal.get("FETCH_SIZE_factor_nt_16B_reads", float("nan")), cal.get("WRITE_SIZE_factor_nt_12B_stores", float("nan")), cal.get("WRITE_SIZE_factor_plain_12B_stores", float("nan"))))
        out.append("")
    sj = os.path.join(PROF, "%s_autotune_stability.json" % TAG)
    if os.path.exists(sj):
        st = json.load(open(sj))
        out.append("**Stability of the launch-shape search** (`profiles/%s_autotune_stability.json`: consecutive `python bench.py` runs on one box):" % TAG)
        out.append("")
        out.append("| workload | picks (entry index per run) | kernels | ms per step per run (µs) |")
        out.append("|---|---|---|---|")
        for name, runs in st.items():
            out.append("| %s | %s | %s | %s |" % (name, " ".join(str(r["pick"]) for r in runs), ", ".join(sorted({"`%s`" % r["kernel"] for r in runs})),
                                              " ".join("%.2f" % (r["ms_per_step"] * 1e3) for r in runs)))
        out.append("")
    out.append(END)
    return "\n".join(out)


def build_readme():
    """the short headline table of README.md"""
    out = [BEGIN, "", "| Config (BASELINE.json) | frame (µs) | verts/s | kernel: algorithmic GB/s | % of 8 TB/s (kernel / frame) |", "|---|---|---|---|---|"]
    for suffix, what in LINES:
        if suffix in ("c5_allgather1", "rehearse8", "shard8_auto", "c4_rehearse8", "c5_steps20", "shard8_steps20"):
            continue
        d = load(suffix)
        if d is None:
            continue
        r = d["roofline"]
        out.append("| %s | %s | %.3g | %.0f | %.1f / %.1f |" % (what.split(" — ")[0], us(d["ms_per_step"]), d["value"], r["achieved"], 100 * r["frac"], 100 * r["frame_frac"]))
    d = load("c5")
    if d and d.get("cpu_baseline"):
        out.append("| CPU baseline of the C5 line (%s, %d threads) | — | %.3g | — | — |" % (d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["value"]))
    out += ["", "(generated from `profiles/%s_bench_*.json` by `tools/design_tables.py`; full tables, rocprof averages and PMC traffic: DESIGN.md §7)" % TAG, END]
    return "\n".join(out)


def main():
    rc = 0
    for fname, builder in (("DESIGN.md", build), ("README.md", build_readme)):
        path = os.path.join(ROOT, fname)
        text = open(path).read()
        if BEGIN not in text or END not in text:
            sys.stderr.write("%s has no GENERATED block\n" % fname)
            return 2
        a, b = text.index(BEGIN), text.index(END) + len(END)
        new = builder()
        if "--check" in sys.argv:
            if text[a:b] != new:
                import difflib
                sys.stderr.write("%s's generated tables differ from profiles/ — run python tools/design_tables.py\n" % fname)
                sys.stderr.write("".join(list(difflib.unified_diff(text[a:b].splitlines(True), new.splitlines(True), fname, "profiles/"))[:60]))
                rc = 1
        else:
            open(path, "w").write(text[:a] + new + text[b:])
    return rc


if __name__ == "__main__":
    sys.exit(main())
