#!/bin/bash
# which GPU test fails one run in ten? the timing-dependent ones, repeated, first failure kept
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/flake; rm -rf $O; mkdir -p $O
t0=$(date +%s)
for i in $(seq 1 12); do
  [ $(( $(date +%s) - t0 )) -gt 170 ] && break
  timeout 120 python -m pytest tests/test_gpu_round3.py -k prefetch -x -q -rf > $O/prefetch_$i.txt 2>&1 || { echo "prefetch FAILED in round $i"; grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" $O/prefetch_$i.txt | tail -40; break; }
  echo "prefetch round $i ok ($(( $(date +%s) - t0 )) s)"
done
t1=$(date +%s)
for i in $(seq 1 6); do
  [ $(( $(date +%s) - t1 )) -gt 110 ] && break
  timeout 120 python -m pytest tests/test_bench_gpu.py -k two_rank_run -x -q -rf > $O/bench2_$i.txt 2>&1 || { echo "bench2 FAILED in round $i"; grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" $O/bench2_$i.txt | tail -30 | cut -c1-1500; break; }
  echo "bench2 round $i ok ($(( $(date +%s) - t1 )) s)"
done
