#!/bin/bash
# Profiles for profiles/: for each workload (C5, one 1/2, 1/4 and 1/8 shard of C5, C4, C3, demo) a rocprofv3 kernel-trace + stats pass of the
# bench command and two HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, kernel-trace only —
# MI355X_MICROARCH.md §HBM), plus a calibration pass over membench's known-byte kernels in the same counters.
# Condense with: python tools/parse_prof.py <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
declare -A CFG
CFG[c5]="--steps 40 --warmup 5"
CFG[shard]="--verts 125184 --steps 200 --warmup 20"
CFG[shard2]="--verts 500224 --steps 80 --warmup 10"
CFG[shard4]="--verts 250112 --steps 150 --warmup 20"
CFG[c4]="--config c4 --steps 100 --warmup 10"
CFG[c3]="--config c3 --steps 200 --warmup 20"
CFG[demo]="--config demo --steps 300 --warmup 20"
for c in c5 shard2 shard4 shard c4 c3 demo; do
  B="python $R/bench.py ${CFG[$c]} --no-cpu-baseline --no-sampled-loop --frames-in-flight 1 --no-pair-loop"
  echo "== $c: kernel trace + stats"
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -o bench -- $B > $O/trace_$c.log 2>&1 || echo "FAILED trace $c"
  grep '^{' $O/trace_$c.log | tail -1 > $O/line_$c.json
  echo "== $c: pmc FETCH_SIZE / WRITE_SIZE"
  timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$c -o bench -- $B > $O/fetch_$c.log 2>&1 || echo "FAILED fetch $c"
  timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$c -o bench -- $B > $O/write_$c.log 2>&1 || echo "FAILED write $c"
done
echo "== calibration (membench quick) FETCH_SIZE / WRITE_SIZE"
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o mb -- $R/tools/membench quick > $O/cal_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o mb -- $R/tools/membench quick > $O/cal_write.log 2>&1
cd $R; find gpurun_out/prof -name "*.csv" | wc -l; du -sh gpurun_out/prof
