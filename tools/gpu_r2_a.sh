#!/bin/bash
# round 2, call A: full GPU test-suite, LDS / store micro-benchmarks, C4 shape sweep, C4 counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r2a; O=$R/gpurun_out/r2a
cd $R
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
echo "== ldsbench"
timeout 200 tools/ldsbench 2>&1 | tee $O/ldsbench.txt
echo "== c4 sweep"
timeout 400 python tools/c4_sweep.py 2>&1 | tee $O/c4_sweep.txt
echo "== c4 counters (heuristic plan)"
cd /tmp && export TMPDIR=/tmp
C4="python $R/bench.py --config c4 --steps 30 --warmup 3 --no-cpu-baseline --no-autotune"
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
G3="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
G4="FETCH_SIZE"
G5="WRITE_SIZE"
for g in 1 2 3 4 5; do
  eval "PM=\$G$g"
  timeout 300 rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/c4_g$g -o p -- $C4 > $O/c4_g$g.log 2>&1 || echo "FAILED g$g: $(tail -2 $O/c4_g$g.log)"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_trace -o p -- $C4 > $O/c4_trace.log 2>&1
cd $R
python3 - <<'P'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/r2a/c4_g*/')):
    f = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0][-60:]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if 'rz_' in k: print(os.path.basename(d.rstrip('/')), k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, 'n=%d' % len(next(iter(v.values()))))
for f in glob.glob('gpurun_out/r2a/c4_trace/**/*kernel_stats.csv', recursive=True):
    print(open(f).read()[:1500])
P
du -sh $O; find $O -name "*.csv" | wc -l
