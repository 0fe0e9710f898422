R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c5cnt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for c in c5 shard; do
  A=""; [ $c = shard ] && A="--verts 125184"
  for g in 1 2; do
    eval "PM=\$G$g"
    timeout 300 rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/${c}_g$g -o p -- python $R/bench.py $A --steps 30 --warmup 3 --no-cpu-baseline --no-autotune --no-sampled-loop --clock-warm-seconds 0.2 > $O/${c}_g$g.log 2>&1 || echo FAILED
  done
done
cd $R
python3 - <<'P'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/c5cnt/*_g*/')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d + 'p_counter_collection.csv')):
        k = r['Kernel_Name']
        if 'rz_deform_' in k: agg[k.split('::')[-1].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(os.path.basename(d.rstrip('/')), k, " ".join("%s=%.0f" % (c, sum(x) / len(x)) for c, x in sorted(v.items())), "n=%d" % len(next(iter(v.values()))))
P
rm -rf $O
