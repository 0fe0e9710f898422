#!/bin/bash
# Round 4, last evidence session on the final tree: the tracked bench lines (gpu_full.sh nobench) and the per-wave timelines
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/gpu_full.sh nobench 2>&1 | grep -E "^bench_|^stab_" | cut -c1-200
O=gpurun_out/ev3; rm -rf $O; mkdir -p $O
for c in c2 c3 sparse2 demo sampled-c2 local-c2 sampled-demo shard c4 c5; do
  timeout 200 python tools/timeline.py $c 2>&1 | grep -v "amdgpu.ids" > $O/timeline_$c.txt; head -2 $O/timeline_$c.txt | tail -1 | cut -c1-160
done
