#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4q; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "crowd or instance or c4 or subset or fuzz or largest or fk" 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -4 | tee $O/pytest_subset.txt
timeout 200 python tools/timeline.py c4 2>&1 | grep -v "amdgpu.ids\|per XCD\|late wave" | tee $O/timeline_c4.txt
for rep in 1 2 3; do
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py c4 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py c4 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
