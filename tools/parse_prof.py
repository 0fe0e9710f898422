#!/usr/bin/env python3
"""Condense gpurun_out/prof/ (rocprofv3 CSVs written by tools/gpu_profile.sh) into the tracked
summaries under profiles/: kernel-trace stats tables and the PMC-derived HBM traffic
(profiles/pmc_traffic.json, read back by bench.py for `roofline.traffic`).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are
collected in separate passes, are in KiB, and on gfx950 FETCH_SIZE counts exactly half of a wide
coalesced read stream — the factor is re-derived here from membench's known-byte kernels run
under the same counters (calibration pass) instead of being assumed."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"


def counters(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def find(agg, name_part, counter):
    for (k, c), v in agg.items():
        if name_part in k and c == counter:
            return v
    return None


def stats_table(path, title, out):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write("# %s\n# rocprofv3 --kernel-trace --stats --output-format csv (MI355X, gfx950); times in us\n" % title)
        f.write("%-100s %7s %12s %10s %10s %10s %6s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
        for r in rows:
            f.write("%-100s %7s %12.1f %10.3f %10.3f %10.3f %6s\n" % (
                r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    return rows


os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
stats_table(os.path.join(P, "trace", "bench_kernel_stats.csv"), "python bench.py --steps 40 --warmup 5 (C5: 1M verts / 256 bones / 64 morphs, 1 GPU)",
            os.path.join(ROOT, "profiles", "%s_kernel_stats_c5.txt" % tag))
for name, title in (("trace_shard", "python bench.py --verts 125952 (one 1/8 shard of C5)"),
                    ("trace_c4", "python bench.py --verts 30000 --bones 200 --morphs 0 --instances 256 (C4)")):
    p = os.path.join(P, name, "bench_kernel_stats.csv")
    if os.path.exists(p):
        stats_table(p, title, os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.txt" % (tag, name.split("_")[1])))

cal_f = counters(os.path.join(P, "cal_fetch", "mb_counter_collection.csv"))
cal_w = counters(os.path.join(P, "cal_write", "mb_counter_collection.csv"))
known_kib = 828 * 1024.0
f_read = known_kib / find(cal_f, "k_read<true, 4>", "FETCH_SIZE")         # nontemporal 16 B/lane stream
f_read_plain = known_kib / find(cal_f, "k_read<false, 4>", "FETCH_SIZE")
f_write3 = (known_kib * 1024 // 12 * 12 / 1024.0) / find(cal_w, "k_fill3<true>", "WRITE_SIZE")   # NT 12 B/lane stores
f_write4 = known_kib / find(cal_w, "k_fill<false>", "WRITE_SIZE")
fe = counters(os.path.join(P, "fetch", "bench_counter_collection.csv"))
wr = counters(os.path.join(P, "write", "bench_counter_collection.csv"))
fetch_kib = find(fe, "rz_deform_kernel", "FETCH_SIZE")
write_kib = find(wr, "rz_deform_kernel", "WRITE_SIZE")
read_b = fetch_kib * 1024 * f_read
write_b = write_kib * 1024 * f_write3
V, B, M, I = 1000000, 256, 64, 1
alg_read = V * (36 + 12 * M) + B * 128 + M * 4
alg_write = V * 24
rec = {
    "V%d_B%d_M%d_I%d" % (V, B, M, I): {
        "hbm_bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
        "raw_FETCH_SIZE_KiB": fetch_kib, "raw_WRITE_SIZE_KiB": write_kib,
        "calibration": {"FETCH_SIZE_factor_nt_16B_reads": f_read, "FETCH_SIZE_factor_plain_16B_reads": f_read_plain,
                        "WRITE_SIZE_factor_nt_12B_stores": f_write3, "WRITE_SIZE_factor_16B_stores": f_write4,
                        "known_bytes": "tools/membench quick: 828 MiB read / filled per launch"},
        "algorithmic_read_bytes": alg_read, "algorithmic_write_bytes": alg_write,
        "traffic_over_algorithmic": (read_b + write_b) / (alg_read + alg_write),
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 40 --warmup 5",
    }
}
json.dump(rec, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.txt" % tag), "w") as f:
    f.write(json.dumps(rec, indent=1) + "\n")
print(json.dumps(rec, indent=1))
