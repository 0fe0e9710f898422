#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r4o; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "sampl or fk or FK or anim or motion or bone_morph or doubling or local_poses or fuzz or physics or override" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 | tee $O/pytest_subset.txt
for c in sampled-c2 local-c2 sampled-demo; do timeout 200 python tools/timeline.py $c 2>&1 | grep -v "amdgpu.ids\|per XCD\|late wave" | tee -a $O/timeline.txt; done
for rep in 1 2; do
REZE_LIB=$R/tools/_tmp/old/libreze_deform_old.so timeout 300 python tools/ab_r4.py anim 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 300 python tools/ab_r4.py anim 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
