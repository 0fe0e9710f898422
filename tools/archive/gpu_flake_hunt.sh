#!/bin/bash
# Is the GPU suite flaky? The whole -m gpu suite N times back to back on one box (default 5), counts kept -> gpurun_out/flake/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
N=${1:-5}
O=gpurun_out/flake; rm -rf $O; mkdir -p $O
echo "# python -m pytest tests -m gpu -q, $N runs back to back on one MI355X box ($(date -u +%Y-%m-%dT%H:%MZ))" > $O/summary.txt
for i in $(seq 1 $N); do
  t0=$(date +%s)
  timeout 1500 python -m pytest tests -m gpu -q -rf > $O/run_$i.txt 2>&1; rc=$?
  line=$(grep -E "passed|failed" $O/run_$i.txt | tail -1)
  echo "run $i: rc=$rc  $line  ($(( $(date +%s) - t0 )) s wall)" | tee -a $O/summary.txt
  if [ $rc -ne 0 ]; then grep -E "^FAILED|^ERROR" $O/run_$i.txt | head -10 | tee -a $O/summary.txt; fi
  grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" $O/run_$i.txt | tail -30 > $O/run_${i}_tail.txt; rm -f $O/run_$i.txt
done
