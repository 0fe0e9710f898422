#!/bin/bash
# SQ / LDS / TCC / TCP counters of the dominant kernel of every tracked workload on the CURRENT tree (round 5: replaces gpu_c4_counters.sh,
# gpu_c5_counters.sh, gpu_pmc_sq.sh, gpu_sparse_counters.sh). One PMC group per rocprofv3 run (kernel-trace only, as gpurun demands);
# FETCH_SIZE / WRITE_SIZE traffic comes from gpu_profile.sh. The write-path groups are also collected for tools/archive/storebench — the same
# 184 MB written with no loads and no math — so the C4 kernel's store stalls can be read against the fill's own.
#   usage: [PMC_GROUPS="1 2 3"] tools/archive/gpu_counters.sh [workload ...]      default: c5 shard c4 c4fk demo store, all five groups      -> gpurun_out/counters/summary_<workload>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/counters; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-autotune --no-sampled-loop --frames-in-flight 1 --no-pair-loop --clock-warm-seconds 0.3"
declare -A CMD
CMD[c5]="python $R/bench.py --steps 20 --warmup 3 $COMMON"
CMD[shard]="python $R/bench.py --verts 125184 --steps 60 --warmup 5 $COMMON"
CMD[c4]="python $R/bench.py --config c4 --steps 40 --warmup 4 $COMMON"
CMD[c4fk]="python $R/bench.py --config c4 --device-fk --steps 40 --warmup 4 $COMMON"
CMD[demo]="python $R/bench.py --config demo --steps 100 --warmup 10 $COMMON"
CMD[store]="$R/tools/archive/storebench"
declare -A KEY
KEY[c5]="rz_deform_dense"; KEY[shard]="rz_deform_dense"; KEY[c4]="rz_skin_instances_kernel"; KEY[c4fk]="rz_skin_instances_fk"; KEY[demo]="rz_deform_small"; KEY[store]="void k<0>"
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
G3="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
# (Not collected: the TCC_EA0_WRREQ* / TCC_WRITE / TCC_TAG_STALL / TCC_BUSY groups. On this pool rocprofv3 never finishes a run that asks for
# them — "There are 1 incomplete dispatches" until the timeout, 5 minutes of GPU time per attempt, round 5 — so the write path is read at
# the L1's side of it: requests, their summed latency and the cycles the L1 stalls on pending ones.)
G4="SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"
G5="TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_REQUEST_sum"
WL=${@:-c5 shard c4 c4fk demo store}
for c in $WL; do
  groups=${PMC_GROUPS:-"1 2 3 4 5"}; [ $c = store ] && groups="1 4 5"
  for g in $groups; do
    eval "PM=\$G$g"
    timeout 150 rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/${c}_g$g -o p -- ${CMD[$c]} > $O/${c}_g$g.log 2>&1 || echo "FAILED $c g$g: $(tail -1 $O/${c}_g$g.log)"
  done
  KEYSTR="${KEY[$c]}" WLNAME=$c python3 - <<'P' | tee $O/summary_$c.txt
import csv, glob, collections, os
key, wl = os.environ["KEYSTR"], os.environ["WLNAME"]
out = collections.OrderedDict()
for d in sorted(glob.glob('%s_g*/' % os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/counters", wl))):
    f = glob.glob(d + '*counter_collection.csv')
    if not f:
        print("%s %s: no counter file" % (wl, os.path.basename(d.rstrip('/')))); continue
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if key in r['Kernel_Name']:
            per_kernel[r['Kernel_Name'].split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    if not per_kernel:
        print("%s %s: no launch of a kernel named *%s*" % (wl, os.path.basename(d.rstrip('/')), key)); continue
    name, cnt = max(per_kernel.items(), key=lambda kv: len(next(iter(kv[1].values()))))      # the most-launched shape
    print("%s %s %s (n=%d): %s" % (wl, os.path.basename(d.rstrip('/')).split('_')[-1], name, len(next(iter(cnt.values()))), " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cnt.items()))))
P
  rm -rf $O/${c}_g?
done
