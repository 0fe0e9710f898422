#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4p; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -5 | tee $O/pytest_gpu.txt
for c in c2 demo shard; do timeout 200 python tools/timeline.py $c 2>&1 | grep -v "amdgpu.ids\|per XCD\|late wave" | tee -a $O/timeline.txt; done
for rep in 1 2; do
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 400 python tools/ab_r4.py small dense 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 400 python tools/ab_r4.py small dense 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
