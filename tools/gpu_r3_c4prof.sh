#!/bin/bash
# C4 evidence of round 3: kernel trace + stats of the bench command, SQ / LDS / TCC counters of the default plan's kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3c4; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-sampled-loop --frames-in-flight 1 --no-pair-loop"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $B > $O/trace.log 2>&1 || echo "FAILED trace"
grep '^{' $O/trace.log | tail -1 > $O/line_under_rocprof.json
cd $R
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_c4.json
bash tools/gpu_c4_counters.sh > $O/counters.log 2>&1
cp gpurun_out/c4cnt/summary.txt $O/sq_counters.txt
python - <<'P'
import csv, json
rows = list(csv.DictReader(open('gpurun_out/r3c4/trace/bench_kernel_stats.csv')))
for r in rows[:8]:
    print("%-90s calls %6s avg %9.3f us min %9.3f max %9.3f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
d = json.load(open('gpurun_out/r3c4/bench_c4.json'))
print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["frame_frac"], d["config"]["autotune_pick"], d["config"]["grid"], d["config"]["inst_group"], d["roofline"]["kernel_ms_check"])
P
cat $O/sq_counters.txt
rm -rf $O/trace/*/*_agent_info.csv
