#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4j; rm -rf $O; mkdir -p $O
timeout 200 python tools/timeline.py c4 2>&1 | grep -v "amdgpu.ids\|per XCD" | tee -a $O/timeline.txt
for rep in 1 2 3; do
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py c4 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py c4 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
