#!/usr/bin/env python3
"""Condense gpurun_out/prof/ (rocprofv3 CSVs written by tools/gpu_profile.sh) into the tracked summaries under profiles/:
kernel-trace stats tables per workload and the PMC-derived HBM traffic per launch (profiles/pmc_traffic.json, keyed "<shape>|<kernel>";
bench.py looks `roofline.traffic` up by the kernel its plan actually launches — and carries no traffic figure when that
kernel was never measured; it says which in `roofline.traffic_source`).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in separate passes, are
in KiB, and on gfx950 FETCH_SIZE counts exactly half of a wide coalesced read stream — the factor is re-derived here from
membench's known-byte kernels run under the same counters (calibration pass) instead of being assumed. The dominant
kernel of a workload = the rz_* deform / skin kernel with the most launches in that run (the variant rz_autotune kept)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"

WORK = {   # name -> (title, V per GPU, B, M, I)
    "c5": ("python bench.py --steps 40 --warmup 5 (C5: 1M verts / 256 bones / 64 morphs, 1 GPU)", 1000000, 256, 64, 1),
    "shard2": ("python bench.py --verts 500224 (one 1/2 shard of C5)", 500224, 256, 64, 1),
    "shard4": ("python bench.py --verts 250112 (one 1/4 shard of C5)", 250112, 256, 64, 1),
    "shard": ("python bench.py --verts 125184 (one 1/8 shard of C5)", 125184, 256, 64, 1),
    "c4": ("python bench.py --config c4 (256 x 30000 verts / 200 bones, instanced)", 30000, 200, 0, 256),
    "c3": ("python bench.py --config c3 (30000 verts / 200 bones / 64 morphs)", 30000, 200, 64, 1),
    "demo": ("python bench.py --config demo (28842 verts / 349 bones / 60 sparse morphs, the demo model's statistics)", 28842, 349, 60, 1),
}


def counters(path):
    """{kernel name: {counter: (mean, launches)}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in agg.items()}


def dominant(agg, counter):
    best = None
    for k, d in agg.items():
        if ("rz_deform_" in k or "rz_skin_instances" in k) and counter in d:
            if best is None or d[counter][1] > agg[best][counter][1]:
                best = k
    return best


def find(agg, name_part, counter):
    for k, d in agg.items():
        if name_part in k and counter in d:
            return d[counter][0]
    return None


def stats_table(path, title, out, trace=None):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write("# %s\n# rocprofv3 --kernel-trace --stats --output-format csv (MI355X, gfx950); times in us\n" % title)
        f.write("%-100s %7s %12s %10s %10s %10s %6s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
        for r in rows:
            f.write("%-100s %7s %12.1f %10.3f %10.3f %10.3f %6s\n" % (
                r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
        if trace and os.path.exists(trace):
            # The stats above average a kernel NAME over every launch shape the run used (the launch-shape search times a dozen);
            # the kernel trace tells them apart: one row per (kernel, workgroups, threads, LDS bytes). The shape with the most
            # launches is the plan the bench loops ran.
            shapes = collections.defaultdict(list)
            for r in csv.DictReader(open(trace)):
                k = r["Kernel_Name"]
                if "rz_deform_" not in k and "rz_skin_instances" not in k:
                    continue
                wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
                grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
                name = k.replace("void (anonymous namespace)::", "").split("(")[0]
                shapes[(name, grid, wg, int(r["LDS_Block_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            f.write("# by launch shape (kernel trace of the same run): kernel | workgroups | threads | LDS bytes | calls | avg_us | min_us | max_us\n")
            for (name, grid, wg, lds), d in sorted(shapes.items(), key=lambda kv: -len(kv[1])):
                f.write("SHAPE %-70s %6d %5d %7d %7d %10.3f %10.3f %10.3f\n" % (name, grid, wg, lds, len(d), sum(d) / len(d), min(d), max(d)))
    return rows


os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
cal_f = counters(os.path.join(P, "cal_fetch", "mb_counter_collection.csv"))
cal_w = counters(os.path.join(P, "cal_write", "mb_counter_collection.csv"))
known_kib = 828 * 1024.0
f_read = known_kib / find(cal_f, "k_read<true, 4>", "FETCH_SIZE")         # nontemporal 16 B/lane stream
f_read_plain = known_kib / find(cal_f, "k_read<false, 4>", "FETCH_SIZE")
f_write3 = (known_kib * 1024 // 12 * 12 / 1024.0) / find(cal_w, "k_fill3<true>", "WRITE_SIZE")   # NT 12 B/lane stores
f_write3_plain = (known_kib * 1024 // 12 * 12 / 1024.0) / find(cal_w, "k_fill3<false>", "WRITE_SIZE")
f_write4 = known_kib / find(cal_w, "k_fill<false>", "WRITE_SIZE")
rec = {}
for name, (title, V, B, M, I) in WORK.items():
    st = os.path.join(P, "trace_" + name, "bench_kernel_stats.csv")
    if os.path.exists(st):
        stats_table(st, title, os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.txt" % (tag, name)), os.path.join(P, "trace_" + name, "bench_kernel_trace.csv"))
    line = os.path.join(P, "line_%s.json" % name)
    if os.path.exists(line) and os.path.getsize(line) > 2:
        open(os.path.join(ROOT, "profiles", "%s_bench_under_rocprof_%s.json" % (tag, name)), "w").write(open(line).read())
    fp, wp = os.path.join(P, "fetch_" + name, "bench_counter_collection.csv"), os.path.join(P, "write_" + name, "bench_counter_collection.csv")
    if not (os.path.exists(fp) and os.path.exists(wp)):
        print("no PMC passes for", name)
        continue
    fe, wr = counters(fp), counters(wp)
    # one record per (workload shape, kernel): every deform / skin kernel variant the run launched often enough to average
    # (the launch-shape search tries several; bench.py looks up the one its plan ends up with, and only that one)
    shape = "V%d_B%d_M%d_I%d%s" % (V, B, M, I, "_demo" if name == "demo" else "")
    for kf in sorted(fe):
        if not ("rz_deform_" in kf or "rz_skin_instances" in kf) or "FETCH_SIZE" not in fe[kf] or kf not in wr or "WRITE_SIZE" not in wr[kf]:
            continue
        if fe[kf]["FETCH_SIZE"][1] < 30:
            continue
        fetch_kib, write_kib = fe[kf]["FETCH_SIZE"][0], wr[kf]["WRITE_SIZE"][0]
        kname = kf.replace("void (anonymous namespace)::", "").split("(")[0]
        targs = kname.split("<")[1].split(",")
        # position of NTS (nontemporal output stores) in the template list: dense <S, U, NT, NTS, ...>, small <S, MODE, NTS, ...>, crowd <BLOCK, NTS, ...>
        nts = "true" in targs[3 if "rz_deform_dense" in kname else (2 if "rz_deform_small" in kname else 1)]
        read_b = fetch_kib * 1024 * f_read
        write_b = write_kib * 1024 * (f_write3 if nts else f_write3_plain)
        if I > 1:
            alg_read = V * 36 + I * B * 64 + B * 64  # mesh once, world matrices per instance, inverse bind
            alg_write = I * V * 24
        elif name == "demo":
            alg_read = V * (36 + 4) + 36397 * 16 + B * 128 + M * 4      # sparse: + 4 B of row pointer per vertex and the demo shape's 36 397 entries of 16 B
            alg_write = V * 24
        else:
            alg_read = V * (36 + 12 * M) + B * 128 + M * 4
            alg_write = V * 24
        r = {
            "kernel": kname,
            "launches_counted": fe[kf]["FETCH_SIZE"][1],
            "hbm_bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
            "raw_FETCH_SIZE_KiB": fetch_kib, "raw_WRITE_SIZE_KiB": write_kib,
            "algorithmic_read_bytes": alg_read, "algorithmic_write_bytes": alg_write,
            "traffic_over_algorithmic": None if alg_read is None else (read_b + write_b) / (alg_read + alg_write),
            "note": ("crowd frame: the kernel reads the mesh and the world + inverse-bind matrices and writes the output (the whole-palette form also "
                     "writes the palette copy that keeps the skinMatrixBuffer observable; the bone-subset form does not). Its reads are 4-byte loads + "
                     "LDS-DMA served mostly from L2 / Infinity Cache (whose hits FETCH_SIZE counts); the x2 factor calibrated on 16 B/lane streams is applied "
                     "as an UPPER bound (traffic_over_algorithmic), the raw counter gives traffic_over_algorithmic_raw_fetch") if I > 1 else "",
            "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- " + title.split(" (")[0] + " --no-cpu-baseline --no-sampled-loop --frames-in-flight 1 --no-pair-loop",
        }
        if I > 1:
            r["traffic_over_algorithmic_raw_fetch"] = (fetch_kib * 1024 + write_b) / (alg_read + alg_write)
        rec[shape + "|" + kname] = r
rec["_calibration"] = {"FETCH_SIZE_factor_nt_16B_reads": f_read, "FETCH_SIZE_factor_plain_16B_reads": f_read_plain,
                       "WRITE_SIZE_factor_nt_12B_stores": f_write3, "WRITE_SIZE_factor_plain_12B_stores": f_write3_plain,
                       "WRITE_SIZE_factor_16B_stores": f_write4, "known_bytes": "tools/membench quick: 828 MiB read / filled per launch"}
json.dump(rec, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.txt" % tag), "w") as f:
    f.write(json.dumps(rec, indent=1) + "\n")
print(json.dumps(rec, indent=1))
