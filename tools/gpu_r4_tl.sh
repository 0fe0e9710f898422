#!/bin/bash
# Round 4: per-wave timelines (tools/timeline.py, ablate build) of the small frames, the device-animated ones, a 1/8 shard and C4
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/${1:-r4tl}; mkdir -p $O
for c in c2 sparse2 demo sampled-c2 local-c2 sampled-demo shard c4 c5; do
  echo "== $c"; timeout 200 python tools/timeline.py $c 2>&1 | grep -v amdgpu.ids | tee -a $O/timeline.txt
done
echo "== A/B small"
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py small 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
