#!/bin/bash
# SQ-side counters for the dominant kernel on C5, the 1/8 shard and C4 (one pass per counter group).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sq
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
declare -A CFG
CFG[c5]="--steps 20 --warmup 3 --no-cpu-baseline"
CFG[shard]="--verts 125184 --steps 50 --warmup 5 --no-cpu-baseline"
CFG[c4]="--verts 30000 --bones 200 --morphs 0 --instances 256 --steps 30 --warmup 3 --no-cpu-baseline"
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
G3="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
for c in c5 shard c4; do
  for g in 1 2 3; do
    eval "PM=\$G$g"
    timeout 300 rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/${c}_g$g -o p -- python $R/bench.py ${CFG[$c]} > $O/${c}_g$g.log 2>&1 || echo "FAILED $c g$g: $(tail -2 $O/${c}_g$g.log)"
  done
done
cd $R
python3 - <<'P'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/sq/*_g*/')):
    f = glob.glob(d + '*counter_collection.csv')
    if not f: print(d, 'no csv'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'rz_deform_' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(d.rstrip('/')), {k: round(sum(v) / len(v), 1) for k, v in agg.items()})
P
