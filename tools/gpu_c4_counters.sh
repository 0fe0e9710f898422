#!/bin/bash
# SQ / LDS / TCC counters of the C4 skin kernel the default plan launches (one PMC group per run; FETCH / WRITE come from gpu_profile.sh).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c4cnt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C4="python $R/bench.py --config c4 --steps 30 --warmup 3 --no-cpu-baseline --no-autotune --no-sampled-loop --frames-in-flight 1 --no-pair-loop --clock-warm-seconds 0.2"
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
G3="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
for g in 1 2 3; do
  eval "PM=\$G$g"
  timeout 300 rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/g$g -o p -- $C4 > $O/g$g.log 2>&1 || echo "FAILED g$g"
done
cd $R
python3 - <<'P'
import csv, glob, collections, os
lines = []
for d in sorted(glob.glob('gpurun_out/c4cnt/g*/')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d + 'p_counter_collection.csv')):
        k = r['Kernel_Name']
        k = k.split('::')[-1].split('(')[0] if 'skin_instances' in k else None
        if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        lines.append("%s %-40s %s (n=%d)" % (os.path.basename(d.rstrip('/')), k, " ".join("%s=%.1f" % (c, sum(x) / len(x)) for c, x in sorted(v.items())), len(next(iter(v.values())))))
open('gpurun_out/c4cnt/summary.txt', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
P
rm -rf $O/g1 $O/g2 $O/g3
