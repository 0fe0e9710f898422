#!/bin/bash
# SQ / LDS / TCC counters of the sparse-morph kernel (rz_deform_kernel<4, 1, 2, ...>) on the demo-shaped frame and on the same
# entry count spread at 2 % — one PMC group per run (round-4 review item 1). Summary -> gpurun_out/spcnt/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/spcnt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
G3="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
for c in demo sparse2; do
  B="python $R/bench.py --config $c --steps 300 --warmup 20 --no-cpu-baseline --no-autotune --no-sampled-loop --frames-in-flight 1 --no-pair-loop --clock-warm-seconds 0.2"
  for g in 1 2 3; do
    eval "PM=\$G$g"
    timeout 300 rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/${c}_g$g -o p -- $B > $O/${c}_g$g.log 2>&1 || echo "FAILED $c g$g"
  done
done
cd $R
python3 - <<'P'
import csv, glob, collections, os
lines = []
for d in sorted(glob.glob('gpurun_out/spcnt/*_g*/')):
    f = glob.glob(d + '*counter_collection.csv')
    if not f:
        lines.append(os.path.basename(d.rstrip('/')) + ' no csv'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'rz_deform_' not in k: continue
        k = k.split('::')[-1].split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        n = len(next(iter(v.values())))
        if n < 100: continue
        lines.append("%s %-52s %s (n=%d)" % (os.path.basename(d.rstrip('/')), k, " ".join("%s=%.1f" % (c, sum(x) / len(x)) for c, x in sorted(v.items())), n))
open('gpurun_out/spcnt/summary.txt', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
P
rm -rf $O/*_g1 $O/*_g2 $O/*_g3
